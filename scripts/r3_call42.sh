#!/bin/bash
# round 3, call 43: wf_trav (BVH worlds) with sequential ifs (168 -> 99 VGPRs) vs the chain (build/variants/nest3)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 2 --warmup 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:50])"; }
for L in build/variants/nest3/lib/libmort_hip.so mort_amd/lib/libmort_hip.so build/variants/nest3/lib/libmort_hip.so mort_amd/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## $L"
  b --mode wave || exit 1; b --scene 10 --width 1200 --spp 100 --mode wave
done
unset MORT_HIP_LIB
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
