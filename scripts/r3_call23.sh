#!/bin/bash
# round 3, call 24: 16 waves per CU at 128 VGPRs (1024-thread groups, 24 spilled registers) vs 12 waves at 153; t_keep with unroll 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:40], d['kernel']['lds_bytes'], d['kernel'].get('vgprs'))"; }
echo "######## main"; b || exit 1
for th in 48,16,24 48,16,32 48,16,40; do echo "== thresholds $th"; MORT_THRESHOLDS=$th b; done
export MORT_HIP_LIB=build/variants/w4/lib/libmort_hip.so
echo "######## w4 768"; b || exit 1
echo "######## w4 1024"; MORT_FAST_BLOCK_SIZE=1024 b || exit 1
for th in 48,16,24 56,16,24 40,16,16; do echo "== w4 1024 thresholds $th"; MORT_FAST_BLOCK_SIZE=1024 MORT_THRESHOLDS=$th b; done
unset MORT_HIP_LIB
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q > $O/r3A_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r3A_pytest.log
