#!/bin/bash
# round 3, call 8: tiles ordered by their longest pixel instead of their sum
cd "$GRAFT_REPO_ROOT" || exit 1
t() { timeout -k 10 300 python scripts/gen_profile.py $1 $2 $3 4 2>&1 | grep "^frame [23]" | cut -c1-100; }
for key in sum max; do export MORT_TILE_KEY=$key; echo "######## MORT_TILE_KEY=$key"
  echo "== headline"; t 1 1200 500
  echo "== s8 800x800x100 block 768 / 512"; MORT_GEN_BLOCK_SIZE=768 MORT_GEN_PRIO_LANES=0 t 8 800 100; MORT_GEN_BLOCK_SIZE=512 MORT_GEN_PRIO_LANES=0 t 8 800 100
  echo "== s8 800x800x1000 block 768"; MORT_GEN_BLOCK_SIZE=768 MORT_GEN_PRIO_LANES=0 t 8 800 1000 | tail -1
  echo "== s8 1920x1080x49 N=1 and rank 0 of 8, block 768"; for n in 1 8; do MORT_GEN_BLOCK_SIZE=768 MORT_GEN_PRIO_LANES=0 timeout -k 10 100 python scripts/tp_one.py 8 1920 49 $n 1.7777778 2>&1 | tail -1 | cut -c1-200; done
  echo "== headline partitions N=2,4,8 rank 0"; for n in 2 4 8; do timeout -k 10 100 python scripts/tp_one.py 1 1200 500 $n 2>&1 | tail -1 | cut -c1-200; done
done
unset MORT_TILE_KEY
timeout -k 10 400 python -m pytest tests/test_gpu_gen.py tests/test_gpu_parity.py -x -q -k "not config5 and not wavefront" > gpurun_out/r3h_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3h_pytest.log
