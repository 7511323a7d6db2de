#!/bin/bash
# round 3, call 23: four-wide tree by the optimal cut (DP) instead of the greedy one; box-step unroll 1 / 2 / 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q > $O/r3z_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r3z_pytest.log
[ $rc -eq 0 ] || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:40], d['kernel']['lds_bytes'])"; }
for L in build/variants/bin2/lib/libmort_hip.so mort_amd/lib/libmort_hip.so build/variants/u1/lib/libmort_hip.so build/variants/u3/lib/libmort_hip.so mort_amd/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## $L"
  b || exit 1; b --mode throughput
done
export MORT_HIP_LIB=build/variants/prof/lib/libmort_hip.so
timeout -k 10 200 python scripts/gen_profile.py 1 1200 500 2 2>&1 | tail -7 | cut -c1-300
unset MORT_HIP_LIB
for th in 48,16,24 48,16,32 48,12,24 44,16,24 52,16,24; do echo "== thresholds $th"; MORT_THRESHOLDS=$th b; done
