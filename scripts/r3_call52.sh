#!/bin/bash
# round 3, call 53: heavy-wave settings at the final scene's full 961 spp
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 2 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'][:22])"; }
export MORT_GEN_BLOCK_SIZE=1024
for hv in 3,2,12,50 3,2,8,50 4,3,12,40 4,3,8,50 2,1,8,60 5,4,12,40 3,2,16,40 4,3,12,50 5,4,10,45 6,5,12,40 4,3,10,35; do echo "== heavy $hv"; MORT_GEN_HEAVY=$hv b --scene 8 --width 800 --spp 1000 || exit 1; done
