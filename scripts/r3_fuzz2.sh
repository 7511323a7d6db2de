#!/bin/bash
# round 3, final binaries (four-wide own tree, leaf records, 1024-thread build): random cameras on the BVH scenes with the launch's own
# workgroup choice, again with the 1024-thread kernel forced, then all ten scenes through the unified tree (MORT_GEN_MIN_PRIMS=0)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 400 python scripts/fuzz_viewpoints.py 500 41 2>&1 | tail -2 || exit 1
MORT_FAST_BLOCK_SIZE=1024 timeout -k 10 400 python scripts/fuzz_viewpoints.py 400 42 2>&1 | tail -2 || exit 1
MORT_GEN_MIN_PRIMS=0 timeout -k 10 350 python scripts/fuzz_viewpoints.py 250 43 all 2>&1 | tail -2
