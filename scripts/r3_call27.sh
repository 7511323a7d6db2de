#!/bin/bash
# round 3, call 28: generic per-pixel kernel at 4 waves per SIMD (128 registers, 9 LDS stack levels, 4 workgroups per CU) vs 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:50], d['kernel']['lds_bytes'], d['kernel'].get('vgprs'))"; }
for L in mort_amd/lib/libmort_hip.so build/variants/g4/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## $L"
  b --scene 6 --width 800 --spp 1000 || exit 1; b --scene 2 --width 1200 --spp 100; b --scene 3 --width 1200 --spp 100; b --scene 4 --width 1200 --spp 100; b --scene 5 --width 1200 --spp 100; b --scene 7 --width 800 --spp 200
done
