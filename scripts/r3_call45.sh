#!/bin/bash
# round 3, call 46: final scene with fewer lanes per wave (MORT_LANE_CAP) -- does a wave with fewer pixels run its rounds faster?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 2 --warmup 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:40])"; }
for fb in 512 1024; do for cap in 64 48 32 24 16 8; do echo "== block $fb, lane cap $cap"; MORT_GEN_BLOCK_SIZE=$fb MORT_LANE_CAP=$cap b --scene 8 --width 800 --spp 100 || exit 1; done; done
