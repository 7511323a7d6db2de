#!/bin/bash
# round 3, call 6: the nested state loop (scheduler + search states inside, shade step outside) -- parity, then frame times
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_gen.py tests/test_gpu_parity.py tests/test_gpu_throughput.py -x -q -k "not config5" > $O/r3f_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/r3f_pytest.log
[ $rc -eq 0 ] || exit 1
t() { timeout -k 10 200 python scripts/gen_profile.py $1 $2 $3 4 2>&1 | grep "^frame [23]" | cut -c1-110; }
for v in base bvhB2 bvhD2; do echo "== headline $v"; MORT_HIP_LIB=build/variants/$v/lib/libmort_hip.so t 1 1200 500; done
echo "== headline A (default)"; t 1 1200 500
echo "== headline base again"; MORT_HIP_LIB=build/variants/base/lib/libmort_hip.so t 1 1200 500
echo "== headline A again"; t 1 1200 500
for th in 48,16,16 40,16,16 48,24,16 56,16,16 48,16,8 48,16,24 32,16,12; do echo "== headline A thresholds $th"; MORT_THRESHOLDS=$th t 1 1200 500 | head -1; done
echo "== s8 1920x1080x49 base 512 / new 512 / new 768 (prio 0)"
for f in "build/variants/base/lib/libmort_hip.so 512" "mort_amd/lib/libmort_hip.so 512" "mort_amd/lib/libmort_hip.so 768"; do set -- $f; MORT_HIP_LIB=$1 MORT_GEN_BLOCK_SIZE=$2 MORT_GEN_PRIO_LANES=0 timeout -k 10 100 python scripts/tp_one.py 8 1920 49 1 1.7777778 2>&1 | tail -1 | cut -c1-200; done
echo "== s8 800x800x100 base / new 512 / new 768 (prio 0)"
MORT_HIP_LIB=build/variants/base/lib/libmort_hip.so t 8 800 100 | head -1; MORT_GEN_BLOCK_SIZE=512 MORT_GEN_PRIO_LANES=0 t 8 800 100 | head -1; MORT_GEN_BLOCK_SIZE=768 MORT_GEN_PRIO_LANES=0 t 8 800 100 | head -1
echo "== s8 4096x4096x4 base / new 768"; sed -i 's/width=width, spp=spp)/width=width, spp=spp, aspect=(1.0 if width == 4096 else None))/' scripts/gen_profile.py
MORT_HIP_LIB=build/variants/base/lib/libmort_hip.so t 8 4096 4 | head -1; MORT_GEN_BLOCK_SIZE=768 MORT_GEN_PRIO_LANES=0 t 8 4096 4 | head -1
echo "== Cornell 800x800x100 on mega_gen base / new 768, and mega_kernel"; MORT_GEN_MIN_PRIMS=0 MORT_HIP_LIB=build/variants/base/lib/libmort_hip.so t 6 800 100 | head -1; MORT_GEN_MIN_PRIMS=0 MORT_GEN_BLOCK_SIZE=768 MORT_GEN_PRIO_LANES=0 t 6 800 100 | head -1; t 6 800 100 | head -1
