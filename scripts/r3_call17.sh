#!/bin/bash
# round 3, calls 17-18: leaf loop computes both roots first; lazy sphere uv
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3s_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3s_pytest.log
[ $rc -eq 0 ] || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:70], d['kernel']['lds_bytes'])"; }
for L in build/variants/prev/lib/libmort_hip.so mort_amd/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## $L"
  b --scene 8 --width 800 --spp 100; b --scene 8 --width 1920 --aspect 1.7777778 --spp 49; b --scene 8 --width 4096 --aspect 1 --spp 4; b --scene 8 --width 4096 --aspect 1 --spp 4 --mode wave
  MORT_GEN_MIN_PRIMS=0 MORT_GEN_BLOCK_SIZE=768 b --scene 6 --width 800 --spp 100; b --scene 6 --width 800 --spp 1000
done
