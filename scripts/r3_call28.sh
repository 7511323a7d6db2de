#!/bin/bash
# round 3, call 29: chain-bound partitions on MORE waves with FEWER lanes each (lane cap), 1024-thread groups (four waves per SIMD)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
t() { timeout -k 10 120 python scripts/tp_one.py 1 1200 500 $1 2>&1 | tail -1 | cut -c1-200; }
for n in 8 4; do
  echo "######## N=$n default"; t $n || exit 1
  for cap in 16 24 32 40 48 64; do
    echo "== N=$n 1024 threads, lane cap $cap"; MORT_FAST_BLOCK_SIZE=1024 MORT_LANE_CAP=$cap t $n
    echo "== N=$n 1024 threads, lane cap $cap, live thresholds"; MORT_FAST_BLOCK_SIZE=1024 MORT_LANE_CAP=$cap MORT_BVH_DRAIN=3 t $n
  done
  for cap in 32 48; do echo "== N=$n 512 threads drain, lane cap $cap"; MORT_FAST_BLOCK_SIZE=512 MORT_CHAIN_BOUND=1 MORT_LANE_CAP=$cap t $n; done
done
