#!/bin/bash
# Round-3 judged evidence in one pass (writes gpurun_out/r3f_* and refreshes profiles/r3 on the box so that the bench lines can quote the
# counter sets taken minutes earlier):  gpurun --timeout 1190 -- 'bash scripts/r3_final.sh'
# two halves (the whole does not fit one 20-minute call): `r3_final.sh sets` = calibration + counter sets, `r3_final.sh lines` = bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; P=profiles/r3; mkdir -p $P
PART=${1:-sets}
if [ "$PART" = sets ]; then
keep() { cp $O/r3f_$1_pmc_hbm.json $P/$1_pmc_hbm.json && cp $O/r3f_$1_pmc_sq_summary.csv $P/$1_pmc_sq_summary.csv && cp $O/r3f_$1_kernel_stats.csv $P/$1_kernel_stats.csv; }
timeout -k 10 200 python scripts/calib.py $O/r3f_calib.json 2> $O/r3f_calib.log > /dev/null || exit 1; cp $O/r3f_calib.json $P/calib.json; tail -3 $O/r3f_calib.log
bash scripts/profile_set.sh r3f_headline 1 1200 500 mega 0 stats,hbm,sq || exit 1; keep headline
echo "== headline set done"
bash scripts/profile_set.sh r3f_c3 6 800 1000 mega 0 stats,hbm,sq || exit 1; keep c3
echo "== c3 set done"
STEPS=2 bash scripts/profile_set.sh r3f_s8 8 800 1000 mega 0 stats,hbm,sq || exit 1; keep s8
echo "== s8 set done"
STEPS=3 bash scripts/profile_set.sh r3f_wave_s1 1 1200 500 wave 0 stats,hbm,sq || exit 1; keep wave_s1
echo "== wave_s1 set done"
STEPS=2 bash scripts/profile_set.sh r3f_c5geom_wave 8 4096 4 wave 1 stats,hbm,sq || exit 1; keep c5geom_wave
echo "== c5geom_wave set done"
STEPS=2 bash scripts/profile_set.sh r3f_c5geom_mega 8 4096 4 mega 1 stats,hbm,sq || exit 1; keep c5geom_mega
echo "== c5geom_mega set done"
exit 0
fi
# ---- lines: the counter sets of the first half must be in profiles/r3 (committed, or copied from gpurun_out/) ----
for k in headline c3 s8 wave_s1 c5geom_wave c5geom_mega; do [ -f $P/${k}_pmc_hbm.json ] || { [ -f $O/r3f_${k}_pmc_hbm.json ] && cp $O/r3f_${k}_pmc_hbm.json $P/${k}_pmc_hbm.json && cp $O/r3f_${k}_pmc_sq_summary.csv $P/${k}_pmc_sq_summary.csv; }; done
timeout -k 10 600 python bench.py > $O/r3f_headline_bench.json 2> $O/r3f_headline_bench.err || exit 1
timeout -k 10 600 python bench.py --mode throughput --cpu-spp 0 > $O/r3f_throughput_bench.json 2> $O/r3f_throughput_bench.err || exit 1
timeout -k 10 600 python bench.py --scene 6 --width 800 --spp 1000 --profile-tag r3/c3 --steps 5 --warmup 1 > $O/r3f_c3_bench.json 2> $O/r3f_c3_bench.err || exit 1
timeout -k 10 600 python bench.py --scene 8 --width 800 --spp 1000 --profile-tag r3/s8 --steps 2 --warmup 1 > $O/r3f_s8_bench.json 2> $O/r3f_s8_bench.err || exit 1
timeout -k 10 600 python bench.py --mode wave --profile-tag r3/wave_s1 --steps 3 --warmup 1 --cpu-spp 0 > $O/r3f_wave_s1_bench.json 2> $O/r3f_wave_s1_bench.err || exit 1
timeout -k 10 300 python bench.py --scene 8 --width 4096 --aspect 1 --spp 4 --profile-tag r3/c5geom_mega --steps 2 --warmup 1 --cpu-spp 0 > $O/r3f_c5geom_mega_bench.json 2> $O/r3f_c5geom.err || exit 1
timeout -k 10 300 python bench.py --scene 8 --width 4096 --aspect 1 --spp 4 --mode wave --profile-tag r3/c5geom_wave --steps 2 --warmup 1 --cpu-spp 0 > $O/r3f_c5geom_wave_bench.json 2>> $O/r3f_c5geom.err || exit 1
timeout -k 10 600 python bench.py --scene 8 --width 800 --spp 100 --mode wave --steps 2 --warmup 1 --cpu-spp 0 > $O/r3f_wave_s8_bench.json 2> $O/r3f_wave_s8_bench.err || exit 1
# config 4's frame at its own spp on one GPU (one step: 4 900 effective spp), and config 5's frame at 49 spp in wavefront mode
timeout -k 10 600 python bench.py --scene 8 --width 1920 --aspect 1.7777778 --spp 5000 --steps 1 --warmup 1 --cpu-spp 0 --no-calib --no-throughput-line > $O/r3f_c4_5000spp_bench.json 2> $O/r3f_c4.err || exit 1
timeout -k 10 300 python scripts/time_partition.py 1 1200 500 mega > $O/r3f_time_partition_s1.json 2> $O/r3f_tp.err || exit 1
timeout -k 10 300 python scripts/time_partition.py 1 1200 500 throughput > $O/r3f_time_partition_s1_throughput.json 2>> $O/r3f_tp.err || exit 1
timeout -k 10 300 python scripts/time_partition.py 8 1920 49 mega 1.7777778 > $O/r3f_time_partition_c4geom_49spp.json 2>> $O/r3f_tp.err || exit 1
for f in headline throughput c3 s8 wave_s1 wave_s8 c5geom_mega c5geom_wave c4_5000spp; do tail -1 $O/r3f_${f}_bench.json | cut -c1-260; done
