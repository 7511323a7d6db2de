#!/bin/bash
# round 3, call 37: a rank of 2 (2 pixels per lane): drain kernels / workgroup shapes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
t() { timeout -k 10 120 python scripts/tp_one.py 1 1200 500 2 2>&1 | tail -1 | cut -c8-90; }
echo "default"; t || exit 1
for fb in 768 512 384; do echo "chain-bound kernels, block $fb"; MORT_FAST_BLOCK_SIZE=$fb MORT_CHAIN_BOUND=1 t; done
for fb in 1024 512; do echo "throughput kernels, block $fb"; MORT_FAST_BLOCK_SIZE=$fb MORT_CHAIN_BOUND=0 t; done
echo "768 drain, spread 6"; MORT_FAST_BLOCK_SIZE=768 MORT_CHAIN_BOUND=1 MORT_SPREAD_SHIFT=6 t
echo "768, live thresholds"; MORT_BVH_DRAIN=3 t
echo "default again"; t
