#!/bin/bash
# round 3, call 5: where the frames end -- per-wave lifetimes (profile build), and which part of the scratch-free mega_bvh costs 16 %
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
P=build/variants/prof/lib/libmort_hip.so
echo "== wave lines: final scene 800x800x100, block 768, no priority pixels"; MORT_HIP_LIB=$P MORT_GEN_BLOCK_SIZE=768 MORT_GEN_PRIO_LANES=0 timeout -k 10 200 python scripts/wave_lines.py 8 800 100 2>&1 | tail -8 | cut -c1-330
echo "== wave lines: final scene 800x800x100, block 512"; MORT_HIP_LIB=$P MORT_GEN_BLOCK_SIZE=512 MORT_GEN_PRIO_LANES=0 timeout -k 10 200 python scripts/wave_lines.py 8 800 100 2>&1 | tail -8 | cut -c1-330
echo "== wave lines: final scene 1920x1080x49, block 768"; MORT_HIP_LIB=$P MORT_GEN_BLOCK_SIZE=768 MORT_GEN_PRIO_LANES=0 timeout -k 10 200 python scripts/wave_lines.py 8 1920 49 1.7777778 2>&1 | tail -8 | cut -c1-330
echo "== wave lines: headline"; MORT_HIP_LIB=$P timeout -k 10 200 python scripts/wave_lines.py 1 1200 500 2>&1 | tail -8 | cut -c1-330
t() { timeout -k 10 200 python scripts/gen_profile.py $1 $2 $3 4 2>&1 | grep "^frame [23]" | cut -c1-110; }
for v in base bvhB bvhC bvhD; do echo "== headline $v"; MORT_HIP_LIB=build/variants/$v/lib/libmort_hip.so t 1 1200 500; done
echo "== headline A (default)"; t 1 1200 500
echo "== headline base again"; MORT_HIP_LIB=build/variants/base/lib/libmort_hip.so t 1 1200 500
echo "== s8 1920x1080x49 base / new 512 / new 768 (prio 0)"
for f in "build/variants/base/lib/libmort_hip.so 512" "mort_amd/lib/libmort_hip.so 512" "mort_amd/lib/libmort_hip.so 768"; do set -- $f; MORT_HIP_LIB=$1 MORT_GEN_BLOCK_SIZE=$2 MORT_GEN_PRIO_LANES=0 timeout -k 10 100 python scripts/tp_one.py 8 1920 49 1 1.7777778 2>&1 | tail -1 | cut -c1-200; done
