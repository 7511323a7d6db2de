#!/bin/bash
# round 3, call 39: inner loop as "if (S) break; if (T) {...} else {L}" (build/variants/nest2) vs "if (T) {...} else if (L) {...} else break"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms', d['roofline']['kernel'], d['kernel'].get('vgprs'))"; }
for L in mort_amd/lib/libmort_hip.so build/variants/nest2/lib/libmort_hip.so mort_amd/lib/libmort_hip.so build/variants/nest2/lib/libmort_hip.so mort_amd/lib/libmort_hip.so build/variants/nest2/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## $L"; b || exit 1; done
