#!/bin/bash
# round 3, call 15: thresholds on the tail-dominated frame; fresh per-wave log
cd "$GRAFT_REPO_ROOT" || exit 1
for fb in 512 768; do echo "== thresholds, s8 800x800x100, block $fb"; MORT_GEN_BLOCK_SIZE=$fb timeout -k 10 500 python scripts/th_sweep.py 8 800 100 MORT_GEN_THRESHOLDS 28,20,4,56 16,12,4,32 8,8,2,16 40,32,8,56 48,20,4,56 28,8,4,56 28,32,4,56 28,20,1,56 28,20,12,56 28,20,4,28 20,20,4,40 36,16,2,56; done
echo "== thresholds, s8 1920x1080x49, block 768"; MORT_GEN_BLOCK_SIZE=768 timeout -k 10 500 python scripts/th_sweep.py 8 1920 49 MORT_GEN_THRESHOLDS 28,20,4,56 16,12,4,32 40,32,8,56 28,8,4,56 28,32,4,56 36,16,2,56 28,20,4,28 2>&1 | sed 's/$/ (1080p)/'
export MORT_HIP_LIB=build/variants/prof/lib/libmort_hip.so
echo "== wave log: final scene 800x800x100, block 512"; MORT_GEN_BLOCK_SIZE=512 timeout -k 10 200 python scripts/wave_lines.py 8 800 100 2>&1 | tail -7 | cut -c1-330
echo "== profile: final scene 1920x1080x10 block 768"; MORT_GEN_BLOCK_SIZE=768 timeout -k 10 200 python scripts/gen_profile.py 8 1920 10 2 2>&1 | sed -n '/^frame 0/,$p' | grep -v "^frame 0" | head -7 | cut -c1-260
