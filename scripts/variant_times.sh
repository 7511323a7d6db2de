#!/bin/bash
# variant_times.sh LIB... : frame times of the three kernel families with each library variant (MORT_HIP_LIB); diagnostic
for L in "$@"; do
  echo "== $L"
  for cfg in "1 1200 500" "6 800 100" "8 800 100"; do
    MORT_HIP_LIB=$L timeout -k 10 200 python scripts/gen_profile.py $cfg 3 2>&1 | grep "^frame 2" | cut -c1-110
  done
done
