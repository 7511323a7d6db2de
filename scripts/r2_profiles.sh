#!/bin/bash
# Round-2 judged profile sets (writes gpurun_out/r2p_*):  gpurun --timeout 1150 -- 'bash scripts/r2_profiles.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/profile_set.sh r2p_headline 1 1200 500 mega 0 bench,stats,hbm,sq || exit 1
bash scripts/profile_set.sh r2p_c3 6 800 1000 mega 0 stats,hbm,sq || exit 1
STEPS=2 bash scripts/profile_set.sh r2p_s8 8 800 1000 mega 0 stats,hbm,sq || exit 1
STEPS=3 bash scripts/profile_set.sh r2p_wave_s1 1 1200 500 wave 0 stats,hbm,sq || exit 1
STEPS=2 bash scripts/profile_set.sh r2p_wave_s8 8 800 100 wave 0 stats || exit 1
timeout -k 10 300 python scripts/time_partition.py > gpurun_out/r2p_time_partition.json 2> gpurun_out/r2p_time_partition.err || exit 1
tail -12 gpurun_out/r2p_time_partition.json
