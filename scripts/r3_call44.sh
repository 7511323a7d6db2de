#!/bin/bash
# round 3, call 45: generic kernel's item loop with sequential ifs (build/variants/seqg) vs the chain
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:50])"; }
for L in mort_amd/lib/libmort_hip.so build/variants/seqg/lib/libmort_hip.so mort_amd/lib/libmort_hip.so build/variants/seqg/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## $L"
  b --scene 6 --width 800 --spp 1000 || exit 1; b --scene 7 --width 800 --spp 200; b --scene 5 --width 1200 --spp 100; b --scene 2 --width 1200 --spp 100
done
