#!/bin/bash
# round 3, call 3: where the scratch-free mega_gen spends its cycles (profile build), and its thresholds at 3 waves per SIMD
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
for fb in 512 768; do echo "== profile build, MORT_GEN_BLOCK_SIZE=$fb"; MORT_GEN_BLOCK_SIZE=$fb MORT_HIP_LIB=build/variants/prof/lib/libmort_hip.so timeout -k 10 200 python scripts/gen_profile.py 8 800 100 2 2>&1 | tail -8 | cut -c1-260; done
echo "== thresholds at 768"; MORT_GEN_BLOCK_SIZE=768 timeout -k 10 500 python scripts/th_sweep.py 8 800 100 MORT_GEN_THRESHOLDS 28,20,4,56 40,24,8,56 48,32,16,56 20,16,4,40 32,32,8,64 56,40,24,60 16,12,2,32 36,20,2,56 28,28,4,28
echo "== thresholds at 512"; MORT_GEN_BLOCK_SIZE=512 timeout -k 10 300 python scripts/th_sweep.py 8 800 100 MORT_GEN_THRESHOLDS 28,20,4,56 40,24,8,56 20,16,4,40 36,20,2,56
echo "== config 5 geometry 4096^2 x 4spp: base / new 512 / new 768"
t() { MORT_HIP_LIB=$1 timeout -k 10 120 python scripts/gen_profile.py 8 4096 4 3 2>&1 | grep "^frame 2" | cut -c1-150; }
sed -i 's/width=width, spp=spp)/width=width, spp=spp, aspect=(1.0 if width == 4096 else None))/' scripts/gen_profile.py
t build/variants/base/lib/libmort_hip.so; MORT_GEN_BLOCK_SIZE=512 t mort_amd/lib/libmort_hip.so; MORT_GEN_BLOCK_SIZE=768 t mort_amd/lib/libmort_hip.so
