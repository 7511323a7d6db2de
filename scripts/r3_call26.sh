#!/bin/bash
# round 3, call 27: unified-tree megakernel 1024 threads vs default at full sizes; Cornell box on the unified tree
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 2 --warmup 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:50], d['kernel']['lds_bytes'], d['kernel'].get('vgprs'))"; }
for bs in default 1024; do echo "######## $bs"; [ $bs = default ] && unset MORT_GEN_BLOCK_SIZE || export MORT_GEN_BLOCK_SIZE=$bs
  b --scene 8 --width 800 --spp 1000 || exit 1; b --scene 8 --width 4096 --aspect 1 --spp 4; b --scene 9 --width 800 --spp 100
  MORT_GEN_MIN_PRIMS=0 b --scene 6 --width 800 --spp 1000; MORT_GEN_MIN_PRIMS=0 b --scene 2 --width 1200 --spp 100
done
unset MORT_GEN_BLOCK_SIZE
echo "######## generic kernel"; b --scene 6 --width 800 --spp 1000
