#!/bin/bash
# round 3, call 49: heavy waves, wider sweep at 100 spp, then the best at the full 961 spp
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 3 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'][:22])"; }
export MORT_GEN_BLOCK_SIZE=1024
for hv in 2,1,8,75 2,1,8,85 2,1,8,65 2,1,6,75 2,1,12,75 3,2,12,50 3,2,12,65 3,2,8,65 3,2,8,75 3,2,16,50 3,1,8,85 4,3,12,50 4,3,8,65 5,3,8,70 5,2,8,80; do echo "== heavy $hv"; MORT_GEN_HEAVY=$hv b --scene 8 --width 800 --spp 100 || exit 1; done
unset MORT_GEN_BLOCK_SIZE
echo "== 961 spp default"; b --scene 8 --width 800 --spp 1000 --steps 2 --warmup 1
export MORT_GEN_BLOCK_SIZE=1024
for hv in 2,1,8,75 3,2,12,50 3,2,8,65; do echo "== 961 spp heavy $hv"; MORT_GEN_HEAVY=$hv b --scene 8 --width 800 --spp 1000 --steps 2 --warmup 1; done
