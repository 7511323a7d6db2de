#!/bin/bash
# round 3, call 48: heavy waves in the unified-tree megakernel (MORT_GEN_HEAVY="mod,num,cap,percent"), final scene 800x800x100
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 3 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'])"; }
MORT_GEN_HEAVY=2,1,16,50 MORT_GEN_BLOCK_SIZE=1024 timeout -k 10 300 python -m pytest tests/test_gpu_gen.py -x -q 2>&1 | tail -2
echo "== default"; b --scene 8 --width 800 --spp 100 || exit 1
for fb in 1024 512; do export MORT_GEN_BLOCK_SIZE=$fb; echo "== block $fb, no heavy waves"; b --scene 8 --width 800 --spp 100
  for hv in 2,1,16,50 2,1,8,50 2,1,8,75 4,1,8,75 4,1,4,75 4,2,16,50 4,3,16,35 3,2,12,50 1,1,16,50; do echo "== block $fb heavy $hv"; MORT_GEN_HEAVY=$hv b --scene 8 --width 800 --spp 100; done
done
