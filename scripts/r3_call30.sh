#!/bin/bash
# round 3, call 31: one rank of 8 (chain-bound, drain kernels): thresholds, workgroup shape, drain flavour, spread
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
t() { timeout -k 10 120 python scripts/tp_one.py 1 1200 500 8 2>&1 | tail -1 | cut -c8-75; }
echo "default"; t || exit 1
for th in 48,16,16 40,16,16 56,16,16 48,12,16 48,20,16 48,16,8 48,16,32 32,16,16 64,16,16 48,8,8; do echo "th $th"; MORT_THRESHOLDS=$th t; done
for fb in 256 512 768; do echo "block $fb"; MORT_FAST_BLOCK_SIZE=$fb MORT_CHAIN_BOUND=1 t; done
for d in 2 3; do echo "drain $d"; MORT_BVH_DRAIN=$d t; done
for sp in 1 2 3 6; do echo "spread $sp"; MORT_SPREAD_SHIFT=$sp t; done
echo "no tile order"; MORT_NO_TILE_ORDER=1 t
echo "tile key sum"; MORT_TILE_KEY=sum t
