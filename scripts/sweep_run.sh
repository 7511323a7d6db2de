#!/bin/bash
# run every variant under build/variants with the CLI: scripts/sweep_run.sh [mort args]
for d in build/variants/*/; do
  n=$(basename $d)
  r=$(LD_LIBRARY_PATH=$d timeout -k 10 120 ./mort_amd/bin/mort "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f ms  %.0f Msamples/s' % (d['seconds']*1e3, d['msamples_per_s']))" 2>&1)
  echo "$n: $r"
done
