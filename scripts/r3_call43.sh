#!/bin/bash
# round 3, call 44: wavefront shade kernels compiled for 4 / 5 / 6 / 8 waves per SIMD
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 2 --warmup 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['config']['workload'][:50])"; }
for L in mort_amd/lib/libmort_hip.so build/variants/ws5/lib/libmort_hip.so build/variants/ws6/lib/libmort_hip.so build/variants/ws8/lib/libmort_hip.so mort_amd/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## $L"
  b --scene 8 --width 4096 --aspect 1 --spp 4 --mode wave || exit 1; b --scene 8 --width 800 --spp 100 --mode wave; b --mode wave
done
