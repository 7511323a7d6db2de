#!/bin/bash
# round 3, call 11: what smaller / bigger leaves of the unified tree do (512-thread workgroups: leaf 3 needs a 120 KB image)
cd "$GRAFT_REPO_ROOT" || exit 1
export MORT_GEN_BLOCK_SIZE=512
for v in leaf3 leaf4 leaf6; do L=build/variants/$v/lib/libmort_hip.so; [ $v = leaf4 ] && L=mort_amd/lib/libmort_hip.so
  echo "######## $v"
  echo "== 1920x1080x49"; MORT_HIP_LIB=$L timeout -k 10 100 python scripts/tp_one.py 8 1920 49 1 1.7777778 2>&1 | tail -1 | cut -c1-160
  echo "== 800x800x100"; MORT_HIP_LIB=$L timeout -k 10 100 python scripts/gen_profile.py 8 800 100 3 2>&1 | grep "^frame 2" | cut -c1-120
done
for v in leaf3p prof leaf6p; do echo "######## profile $v"; MORT_HIP_LIB=build/variants/$v/lib/libmort_hip.so timeout -k 10 200 python scripts/gen_profile.py 8 1920 10 2 2>&1 | sed -n '/^frame 0/,$p' | grep -v "^frame 0" | head -7 | cut -c1-260; done
