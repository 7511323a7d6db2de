#!/bin/bash
# round 3, call 52: heavy waves in the BVH megakernel (MORT_BVH_HEAVY="mod,num,cap,percent"), Scene 1 ranks of 8 / 4 / 2 and the full frame
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
t() { timeout -k 10 120 python scripts/tp_one.py 1 1200 500 $1 2>&1 | tail -1 | cut -c8-70; }
for n in 8 4 2 1; do echo "---- N=$n"; t $n || exit 1
  for hv in 4,1,4,50 4,1,8,50 2,1,8,40 4,1,4,60 8,1,4,60 4,1,2,60 3,1,8,35; do echo "heavy $hv"; MORT_BVH_HEAVY=$hv t $n; done
done
MORT_BVH_HEAVY=4,1,4,50 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
