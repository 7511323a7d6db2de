#!/bin/bash
# round 3, call 4: full GPU suite on the scratch-free kernels (mega_gen + mega_bvh) with priority pixels; frame times
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/r3d_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/r3d_pytest.log
[ $rc -eq 0 ] || exit 1
t() { timeout -k 10 200 python scripts/gen_profile.py $1 $2 $3 3 2>&1 | grep "^frame 2" | cut -c1-120; }
echo "== headline: base / new"; MORT_HIP_LIB=build/variants/base/lib/libmort_hip.so t 1 1200 500; t 1 1200 500
for sh in 0 2 3 4; do echo "== headline new, MORT_SPREAD_SHIFT=$sh"; MORT_SPREAD_SHIFT=$sh t 1 1200 500; done
echo "== Cornell 800x800x1000 mega_kernel base / new"; MORT_HIP_LIB=build/variants/base/lib/libmort_hip.so t 6 800 1000; t 6 800 1000
for fb in 512 768; do for k in 0 1 2 4 8; do echo "== s8 800x800x100 block $fb prio lanes $k"; MORT_GEN_BLOCK_SIZE=$fb MORT_GEN_PRIO_LANES=$k t 8 800 100; done; done
echo "== s8 800x800x1000 base"; MORT_HIP_LIB=build/variants/base/lib/libmort_hip.so t 8 800 1000
for fb in 512 768; do for k in 0 2 4; do echo "== s8 800x800x1000 block $fb prio lanes $k"; MORT_GEN_BLOCK_SIZE=$fb MORT_GEN_PRIO_LANES=$k t 8 800 1000; done; done
for k in 0 2 4; do echo "== config 4 geometry 1920x1080x49, rank 0 of 8, block 768 prio $k"; MORT_GEN_BLOCK_SIZE=768 MORT_GEN_PRIO_LANES=$k timeout -k 10 100 python scripts/tp_one.py 8 1920 49 8 1.7777778 2>&1 | tail -1 | cut -c1-200; done
for k in 0 2; do echo "== config 4 geometry 1920x1080x49, N=1, block 768 prio $k"; MORT_GEN_BLOCK_SIZE=768 MORT_GEN_PRIO_LANES=$k timeout -k 10 100 python scripts/tp_one.py 8 1920 49 1 1.7777778 2>&1 | tail -1 | cut -c1-200; done
