#!/bin/bash
# round 3, call 41: unified-tree megakernel, inner loop as sequential diamonds vs the if / else-if chain (build/variants/prev)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:50], d['kernel'].get('vgprs'))"; }
timeout -k 10 400 python -m pytest tests/test_gpu_gen.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for L in build/variants/prev/lib/libmort_hip.so mort_amd/lib/libmort_hip.so build/variants/prev/lib/libmort_hip.so mort_amd/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## $L"
  b --scene 8 --width 800 --spp 100 || exit 1; b --scene 8 --width 4096 --aspect 1 --spp 4; b --scene 8 --width 1920 --aspect 1.7777778 --spp 49; b --scene 9 --width 800 --spp 100
done
