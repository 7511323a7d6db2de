#!/bin/bash
# round 3, call 34: metal / dielectric hits wait one shade step when at most K lanes of the step carry that material
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:40])"; }
for k in 0 1 2 3 4 6 8 12 0; do echo "== defer_k $k"; MORT_DEFER_K=$k b || exit 1; done
MORT_DEFER_K=3 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
