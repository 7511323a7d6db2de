#!/bin/bash
# round 3, call 2: the scratch-free mega_gen kernel -- parity first, then frame times against the round-2 kernel on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_gen.py tests/test_gpu_throughput.py -x -q -k "not wavefront and not config5" > $O/r3b_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/r3b_pytest.log
[ $rc -eq 0 ] || exit 1
t() { MORT_HIP_LIB=$1 timeout -k 10 120 python scripts/gen_profile.py $2 $3 $4 3 2>&1 | grep "^frame 2" | cut -c1-150; }
echo "== base (round 2 kernel, 512 threads)"; t build/variants/base/lib/libmort_hip.so 8 800 100
echo "== new default"; t mort_amd/lib/libmort_hip.so 8 800 100
for fb in 256 512 768 1024; do echo "== new MORT_GEN_BLOCK_SIZE=$fb"; MORT_GEN_BLOCK_SIZE=$fb t mort_amd/lib/libmort_hip.so 8 800 100; done
echo "== Cornell on mega_gen: base / new 768"; MORT_GEN_MIN_PRIMS=0 t build/variants/base/lib/libmort_hip.so 6 800 100; MORT_GEN_MIN_PRIMS=0 MORT_GEN_BLOCK_SIZE=768 t mort_amd/lib/libmort_hip.so 6 800 100
echo "== Cornell mega_kernel"; t mort_amd/lib/libmort_hip.so 6 800 100
timeout -k 10 100 python scripts/calib.py $O/r3b_calib.json 2> $O/r3b_calib.log > /dev/null; echo "calib rc=$?"; cat $O/r3b_calib.log
