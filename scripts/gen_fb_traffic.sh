#!/bin/bash
# scratch traffic of the unified-tree megakernel by workgroup size (gpurun_out/r2h_*)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
for fb in 768 512; do
  export MORT_GEN_BLOCK_SIZE=$fb
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/r2h_${fb}_$c -- ./mort_amd/bin/mort 8 --width 800 --spp 100 > $O/r2h_${fb}_$c.log 2>&1 || { tail -3 $O/r2h_${fb}_$c.log; exit 1; }
  done
  python3 scripts/pmc_summary.py $O/r2h_${fb}_FETCH_SIZE $O/r2h_${fb}_WRITE_SIZE | grep -v "^kernel" | awk -F, -v fb=$fb '{print fb, $1, $2, $6/1e6 " GB"}'
  tail -1 $O/r2h_${fb}_WRITE_SIZE.log | cut -c1-260
  rm -rf $O/r2h_${fb}_FETCH_SIZE $O/r2h_${fb}_WRITE_SIZE
done
