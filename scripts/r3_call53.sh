#!/bin/bash
# round 3, call 55: very few lanes for a small head (the heaviest pixels only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 3 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'][:22])"; }
export MORT_GEN_BLOCK_SIZE=1024
for hv in 3,2,12,50 2,1,4,75 2,1,4,80 2,1,4,70 3,2,4,70 3,2,6,65 3,2,4,75 4,3,4,70 3,2,5,70; do echo "== heavy $hv"; MORT_GEN_HEAVY=$hv b --scene 8 --width 800 --spp 100 || exit 1; done
