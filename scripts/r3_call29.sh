#!/bin/bash
# round 3, call 30: box tests with v_pk_fma_f32 (build/variants/pk) vs plain fma -- chain-bound ranks and the full frame
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
t() { timeout -k 10 120 python scripts/tp_one.py 1 1200 500 $1 2>&1 | tail -1 | cut -c1-150; }
for L in mort_amd/lib/libmort_hip.so build/variants/pk/lib/libmort_hip.so mort_amd/lib/libmort_hip.so build/variants/pk/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## $L"
  t 8 || exit 1; t 4; t 2; t 1
done
export MORT_HIP_LIB=build/variants/pk/lib/libmort_hip.so
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
