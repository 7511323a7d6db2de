#!/bin/bash
# round 3, call 51: heavy waves as the default (device-side chain-bound decision) vs MORT_GEN_NO_HEAVY=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 3 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'][:22], d['config']['workload'][:44])"; }
run() { echo "---- $*"; MORT_GEN_NO_HEAVY=1 b "$@"; b "$@"; }
run --scene 8 --width 800 --spp 100 || exit 1
run --scene 8 --width 1920 --aspect 1.7777778 --spp 49
run --scene 9 --width 800 --spp 100
run --scene 8 --width 800 --spp 16
run --scene 8 --width 400 --spp 100
run --scene 8 --width 4096 --aspect 1 --spp 4
run --scene 8 --width 800 --spp 1000 --steps 2 --warmup 1
t() { timeout -k 10 120 python scripts/tp_one.py 8 800 100 $1 2>&1 | tail -1 | cut -c8-100; }
for n in 2 8; do echo "---- rank 0 of $n, 800x800x100"; MORT_GEN_NO_HEAVY=1 t $n; t $n; done
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
