#!/bin/bash
# Round-2 judged evidence in one pass (writes gpurun_out/r2f_* and refreshes profiles/r2 on the box so that the bench lines can quote the
# counter sets taken minutes earlier):  gpurun --timeout 1190 -- 'bash scripts/r2_final.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; P=profiles/r2; mkdir -p $P
keep() { cp $O/r2f_$1_pmc_hbm.json $P/$1_pmc_hbm.json && cp $O/r2f_$1_pmc_sq_summary.csv $P/$1_pmc_sq_summary.csv && cp $O/r2f_$1_kernel_stats.csv $P/$1_kernel_stats.csv; }
bash scripts/profile_set.sh r2f_headline 1 1200 500 mega 0 stats,hbm,sq || exit 1; keep headline
echo "== headline set done"
bash scripts/profile_set.sh r2f_c3 6 800 1000 mega 0 stats,hbm,sq || exit 1; keep c3
echo "== c3 set done"
STEPS=2 bash scripts/profile_set.sh r2f_s8 8 800 1000 mega 0 stats,hbm,sq || exit 1; keep s8
echo "== s8 set done"
STEPS=3 bash scripts/profile_set.sh r2f_wave_s1 1 1200 500 wave 0 stats,hbm,sq || exit 1; keep wave_s1
echo "== wave_s1 set done"
STEPS=2 bash scripts/profile_set.sh r2f_wave_s8 8 800 100 wave 0 stats || exit 1; cp $O/r2f_wave_s8_kernel_stats.csv $P/wave_s8_kernel_stats.csv
timeout -k 10 600 python bench.py > $O/r2f_headline_bench.json 2> $O/r2f_headline_bench.err || exit 1
timeout -k 10 600 python bench.py --mode throughput --cpu-spp 0 > $O/r2f_throughput_bench.json 2> $O/r2f_throughput_bench.err || exit 1
timeout -k 10 600 python bench.py --scene 6 --width 800 --spp 1000 --profile-tag r2/c3 --steps 5 --warmup 1 > $O/r2f_c3_bench.json 2> $O/r2f_c3_bench.err || exit 1
timeout -k 10 600 python bench.py --scene 8 --width 800 --spp 1000 --profile-tag r2/s8 --steps 2 --warmup 1 > $O/r2f_s8_bench.json 2> $O/r2f_s8_bench.err || exit 1
timeout -k 10 600 python bench.py --mode wave --profile-tag r2/wave_s1 --steps 3 --warmup 1 --cpu-spp 0 > $O/r2f_wave_s1_bench.json 2> $O/r2f_wave_s1_bench.err || exit 1
timeout -k 10 600 python bench.py --scene 8 --width 800 --spp 100 --mode wave --steps 2 --warmup 1 --cpu-spp 0 > $O/r2f_wave_s8_bench.json 2> $O/r2f_wave_s8_bench.err || exit 1
timeout -k 10 300 python bench.py --scene 8 --width 4096 --aspect 1 --spp 4 --steps 2 --warmup 1 --cpu-spp 0 > $O/r2f_c5geom_mega_bench.json 2> $O/r2f_c5geom.err || exit 1
timeout -k 10 300 python bench.py --scene 8 --width 4096 --aspect 1 --spp 4 --mode wave --steps 2 --warmup 1 --cpu-spp 0 > $O/r2f_c5geom_wave_bench.json 2>> $O/r2f_c5geom.err || exit 1
timeout -k 10 300 python scripts/time_partition.py 1 1200 500 mega > $O/r2f_time_partition_s1.json 2> $O/r2f_tp.err || exit 1
timeout -k 10 300 python scripts/time_partition.py 1 1200 500 throughput > $O/r2f_time_partition_s1_throughput.json 2>> $O/r2f_tp.err || exit 1
for f in headline throughput c3 s8 wave_s1 wave_s8 c5geom_mega c5geom_wave; do tail -1 $O/r2f_${f}_bench.json | cut -c1-260; done
