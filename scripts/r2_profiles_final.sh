#!/bin/bash
# Round-2 final pass: final-scene set with the default (512-thread) unified-tree kernel, then the bench lines that quote the
# committed counter sets (gpurun_out/r2q_*):  gpurun --timeout 1150 -- 'bash scripts/r2_profiles_final.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
STEPS=2 bash scripts/profile_set.sh r2q_s8 8 800 1000 mega 0 stats,hbm,sq || exit 1
mkdir -p profiles/r2 && cp $O/r2q_s8_pmc_hbm.json profiles/r2/s8_pmc_hbm.json && cp $O/r2q_s8_pmc_sq_summary.csv profiles/r2/s8_pmc_sq_summary.csv
timeout -k 10 600 python bench.py > $O/r2q_headline_bench.json 2> $O/r2q_headline_bench.err || exit 1
timeout -k 10 600 python bench.py --scene 6 --width 800 --spp 1000 --profile-tag r2/c3 --steps 5 --warmup 1 > $O/r2q_c3_bench.json 2> $O/r2q_c3_bench.err || exit 1
timeout -k 10 600 python bench.py --scene 8 --width 800 --spp 1000 --profile-tag r2/s8 --steps 2 --warmup 1 > $O/r2q_s8_bench.json 2> $O/r2q_s8_bench.err || exit 1
timeout -k 10 600 python bench.py --mode wave --profile-tag r2/wave_s1 --steps 3 --warmup 1 --cpu-spp 0 > $O/r2q_wave_s1_bench.json 2> $O/r2q_wave_s1_bench.err || exit 1
timeout -k 10 600 python bench.py --scene 8 --width 800 --spp 100 --mode wave --steps 2 --warmup 1 --cpu-spp 0 > $O/r2q_wave_s8_bench.json 2> $O/r2q_wave_s8_bench.err || exit 1
for f in headline c3 s8 wave_s1 wave_s8; do tail -1 $O/r2q_${f}_bench.json | cut -c1-330; done
