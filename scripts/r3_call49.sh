#!/bin/bash
# round 3, call 50: heavy waves (3,2,12,50 and 4,3,12,50 at 1024 threads) on other frames of the unified-tree kernel: where do they help, where do they hurt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 3 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'][:22], d['config']['workload'][:44])"; }
run() { echo "---- $*"; unset MORT_GEN_BLOCK_SIZE MORT_GEN_HEAVY; b "$@"; export MORT_GEN_BLOCK_SIZE=1024; b "$@"; MORT_GEN_HEAVY=3,2,12,50 b "$@"; MORT_GEN_HEAVY=4,3,12,50 b "$@"; MORT_GEN_HEAVY=2,1,6,75 b "$@"; }
run --scene 8 --width 1920 --aspect 1.7777778 --spp 49 || exit 1
run --scene 9 --width 800 --spp 100
run --scene 8 --width 800 --spp 16
run --scene 8 --width 400 --spp 100
run --scene 8 --width 4096 --aspect 1 --spp 4
unset MORT_GEN_BLOCK_SIZE MORT_GEN_HEAVY
t() { timeout -k 10 120 python scripts/tp_one.py 8 800 100 $1 2>&1 | tail -1 | cut -c8-100; }
for n in 2 8; do echo "---- rank 0 of $n, 800x800x100"; t $n; MORT_GEN_BLOCK_SIZE=1024 MORT_GEN_HEAVY=3,2,12,50 t $n; done
