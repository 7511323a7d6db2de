#!/bin/bash
# round 3: random cameras on the final binaries -- BVH scenes, then all ten scenes (the 32-byte-node unified tree, MORT_GEN_MIN_PRIMS=0 so every world takes it)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 500 python scripts/fuzz_viewpoints.py 400 31 2>&1 | tail -3
MORT_GEN_MIN_PRIMS=0 timeout -k 10 600 python scripts/fuzz_viewpoints.py 400 32 all 2>&1 | tail -3
