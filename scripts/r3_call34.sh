#!/bin/bash
# round 3, call 35: final scene, 512-thread launches: sphere records of the leaf step from an LDS copy (32 KB) vs from L2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 2 --warmup 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:50], d['kernel']['lds_bytes'], d['kernel'].get('vgprs'))"; }
timeout -k 10 300 python -m pytest tests/test_gpu_gen.py -x -q 2>&1 | tail -2
for v in 1 0 1 0; do [ $v = 1 ] && unset MORT_GEN_NO_SPH_LDS || export MORT_GEN_NO_SPH_LDS=1; echo "######## sphere copy in LDS: $v"
  b --scene 8 --width 800 --spp 100 || exit 1; b --scene 9 --width 800 --spp 100; b --scene 8 --width 800 --spp 1000
done
