#!/bin/bash
# round 3, call 22: pre-scaled child references, byte-offset traversal stack pointer; threshold sweep for the four-wide step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q > $O/r3w_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r3w_pytest.log
[ $rc -eq 0 ] || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:40], d['kernel']['lds_bytes'])"; }
for L in build/variants/bin2/lib/libmort_hip.so mort_amd/lib/libmort_hip.so build/variants/bin2/lib/libmort_hip.so mort_amd/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## $L"
  b || exit 1; b --mode throughput
done
unset MORT_HIP_LIB
for th in 48,16,16 48,16,8 48,16,24 48,24,16 48,8,16 40,16,16 56,16,16 32,16,16 40,24,8 56,24,24 48,32,16; do echo "== thresholds $th"; MORT_THRESHOLDS=$th b; done
for L in mort_amd/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## partition $L"
  timeout -k 10 300 python scripts/time_partition.py 1 1200 500 mega | python -c "import sys,json; d=json.load(sys.stdin); print({k:[round(t['ms'],1) for t in v] for k,v in d['ranks'].items()})" || exit 1
done
