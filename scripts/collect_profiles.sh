#!/bin/bash
# Collect the round's judged profile set on the GPU box: scripts/collect_profiles.sh TAG   (writes gpurun_out/TAG_*)
# run as:  gpurun --timeout 1100 -- 'bash scripts/collect_profiles.sh v6'
set -o pipefail
TAG=${1:-vX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 400 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { tail -5 $O/${TAG}_bench.err; exit 1; }
tail -1 $O/${TAG}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 bench.py --steps 5 --warmup 1 --cpu-spp 0 > $O/${TAG}_stats.log 2>&1 || exit 1
cp $(find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv && head -3 $O/${TAG}_bench_kernel_stats.csv
rm -rf $O/${TAG}_stats
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/${TAG}_pmc_$c -- ./mort_amd/bin/mort 1 --spp 500 > $O/${TAG}_pmc_$c.log 2>&1 || exit 1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/${TAG}_pmc_sq1 -- ./mort_amd/bin/mort 1 --spp 500 > $O/${TAG}_pmc_sq1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SMEM SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_ACTIVE_INST_ANY --output-format csv -d $O/${TAG}_pmc_sq2 -- ./mort_amd/bin/mort 1 --spp 500 > $O/${TAG}_pmc_sq2.log 2>&1 || exit 1
python3 scripts/pmc_summary.py $O/${TAG}_pmc_sq1 $O/${TAG}_pmc_sq2 > $O/${TAG}_pmc_sq_summary.csv
python3 scripts/pmc_summary.py $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE > $O/${TAG}_pmc_hbm_raw.csv
cat $O/${TAG}_pmc_hbm_raw.csv
rm -rf $O/${TAG}_pmc_sq1 $O/${TAG}_pmc_sq2 $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE
