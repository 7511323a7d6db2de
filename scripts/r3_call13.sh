#!/bin/bash
# round 3, call 13: 32-byte quantized nodes, leaves of <= 2 primitives -- parity, then frame times against the 64-byte / leaf-3 build
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_gen.py tests/test_gpu_throughput.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r3m_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3m_pytest.log
[ $rc -eq 0 ] || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:70], d['kernel']['lds_bytes'])"; }
for L in build/variants/leaf3/lib/libmort_hip.so mort_amd/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## $L"
  b --scene 8 --width 800 --spp 100; MORT_GEN_BLOCK_SIZE=768 b --scene 8 --width 800 --spp 100
  b --scene 8 --width 1920 --aspect 1.7777778 --spp 49; MORT_GEN_BLOCK_SIZE=512 b --scene 8 --width 1920 --aspect 1.7777778 --spp 49
  b --scene 8 --width 4096 --aspect 1 --spp 4
  b --scene 8 --width 4096 --aspect 1 --spp 4 --mode wave
  MORT_GEN_MIN_PRIMS=0 MORT_GEN_BLOCK_SIZE=768 b --scene 6 --width 800 --spp 100
done
unset MORT_HIP_LIB
b --scene 8 --width 800 --spp 1000 --steps 2
