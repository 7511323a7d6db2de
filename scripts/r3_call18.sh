#!/bin/bash
# round 3, call 19: four-wide own tree (DNode4) in the BVH megakernel vs the binary nodes (build/variants/bin2)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r3t_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3t_pytest.log
[ $rc -eq 0 ] || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:70], d['kernel']['lds_bytes'])"; }
for L in build/variants/bin2/lib/libmort_hip.so mort_amd/lib/libmort_hip.so build/variants/bin2/lib/libmort_hip.so mort_amd/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## $L"
  b || exit 1; b --mode throughput; b --scene 10 --width 1200 --spp 100
done
for L in build/variants/bin2/lib/libmort_hip.so mort_amd/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## partition $L"
  timeout -k 10 300 python scripts/time_partition.py 1 1200 500 mega | python -c "import sys,json; d=json.load(sys.stdin); print({k:[round(t['ms'],1) for t in v] for k,v in d['ranks'].items()})" || exit 1
done
