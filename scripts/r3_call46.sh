#!/bin/bash
# round 3, call 47: what a round costs in the waves that end last, against the number of lanes per wave (profile build, per-wave log)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MORT_HIP_LIB=build/variants/prof/lib/libmort_hip.so MORT_GEN_BLOCK_SIZE=1024
for cap in 64 32 16 8 4; do echo "######## lane cap $cap"; MORT_LANE_CAP=$cap timeout -k 10 200 python scripts/wave_lines.py 8 800 100 2>&1 | grep "FRAME\|wave end\|last 5\|last 8\|all waves" | cut -c1-260; done
