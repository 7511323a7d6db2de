#!/bin/bash
# round 3, call 9: what a wave does once the pool is empty -- rounds (2) vs follow one lane (1) vs nothing (0)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 400 python -m pytest tests/test_gpu_gen.py tests/test_gpu_parity.py -x -q -k "not config5 and not wavefront" > gpurun_out/r3i_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3i_pytest.log
[ $rc -eq 0 ] || exit 1
t() { timeout -k 10 300 python scripts/gen_profile.py $1 $2 $3 4 2>&1 | grep "^frame [23]" | cut -c1-100; }
for dm in 0 1 2; do for fb in 768 512; do echo "== s8 800x800x100 MORT_GEN_DRAIN=$dm block $fb"; MORT_GEN_DRAIN=$dm MORT_GEN_BLOCK_SIZE=$fb t 8 800 100; done; done
for dm in 0 2; do echo "== s8 800x800x1000 MORT_GEN_DRAIN=$dm block 768"; MORT_GEN_DRAIN=$dm MORT_GEN_BLOCK_SIZE=768 t 8 800 1000 | tail -1; done
for dm in 0 2; do for n in 1 2 8; do echo "== s8 1920x1080x49 N=$n drain $dm"; MORT_GEN_DRAIN=$dm MORT_GEN_BLOCK_SIZE=768 timeout -k 10 100 python scripts/tp_one.py 8 1920 49 $n 1.7777778 2>&1 | tail -1 | cut -c1-160; done; done
echo "== headline N=1 default"; t 1 1200 500
for dm in 1 2; do echo "== headline N=1 MORT_CHAIN_BOUND=1 MORT_BVH_DRAIN=$dm"; MORT_CHAIN_BOUND=1 MORT_BVH_DRAIN=$dm MORT_SPREAD_SHIFT=6 t 1 1200 500; done
for dm in 1 2; do for n in 2 4 8; do echo "== headline N=$n MORT_BVH_DRAIN=$dm (default kernel choice)"; MORT_BVH_DRAIN=$dm timeout -k 10 100 python scripts/tp_one.py 1 1200 500 $n 2>&1 | tail -1 | cut -c1-160; done; done
for n in 2 4; do echo "== headline N=$n MORT_CHAIN_BOUND=1 drain 2, whole tiles"; MORT_CHAIN_BOUND=1 MORT_BVH_DRAIN=2 MORT_SPREAD_SHIFT=6 timeout -k 10 100 python scripts/tp_one.py 1 1200 500 $n 2>&1 | tail -1 | cut -c1-160; done
echo "== Cornell 800x800x1000 mega_kernel"; t 6 800 1000 | tail -1
