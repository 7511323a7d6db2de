#!/bin/bash
# round 3, call 40: inner loop as two sequential diamonds "if (T) {...} if (L) {...}" (build/variants/nest3): no register copies at a common join
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms', d['roofline']['kernel'], d['kernel'].get('vgprs'))"; }
for L in mort_amd/lib/libmort_hip.so build/variants/nest3/lib/libmort_hip.so mort_amd/lib/libmort_hip.so build/variants/nest3/lib/libmort_hip.so mort_amd/lib/libmort_hip.so build/variants/nest3/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## $L"; b || exit 1; done
t() { timeout -k 10 120 python scripts/tp_one.py 1 1200 500 $1 2>&1 | tail -1 | cut -c8-90; }
for L in mort_amd/lib/libmort_hip.so build/variants/nest3/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## $L"; t 2; t 4; t 8; done
export MORT_HIP_LIB=build/variants/nest3/lib/libmort_hip.so
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
