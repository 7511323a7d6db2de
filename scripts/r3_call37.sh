#!/bin/bash
# round 3, call 38: thresholds as shares of the wave's live lanes (MORT_BVH_DRAIN=3) vs fixed counts, throughput kernels, N = 1 / 2; chain-bound ranks N = 4 / 8
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
t() { timeout -k 10 120 python scripts/tp_one.py 1 1200 500 $1 2>&1 | tail -1 | cut -c8-90; }
for rep in 1 2 3; do for n in 1 2 4 8; do echo "N=$n fixed"; t $n || exit 1; echo "N=$n live"; MORT_BVH_DRAIN=3 t $n; done; done
