#!/bin/bash
# One judged profile set on the GPU box (writes gpurun_out/TAG_*; copy what is to be kept into profiles/):
#   scripts/profile_set.sh TAG SCENE WIDTH SPP MODE [ASPECT] [PARTS]
# PARTS is a subset of "bench,stats,hbm,sq" (default: all).  Each rocprofv3 pass profiles the program itself
# (python3 bench.py / ./mort_amd/bin/mort), never a shell or env wrapper; --pmc passes are separate runs.
#   gpurun --timeout 1100 -- 'bash scripts/profile_set.sh r2_c3 6 800 1000 mega'
set -o pipefail
TAG=$1; SCENE=${2:-1}; WIDTH=${3:-1200}; SPP=${4:-500}; MODE=${5:-mega}; ASPECT=${6:-0}; PARTS=${7:-bench,stats,hbm,sq}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
BARGS="--scene $SCENE --width $WIDTH --spp $SPP --mode $MODE"
MARGS="$SCENE --width $WIDTH --spp $SPP --mode $MODE"
if [ "$ASPECT" != "0" ]; then BARGS="$BARGS --aspect $ASPECT"; MARGS="$MARGS --aspect $ASPECT"; fi
STEPS=${STEPS:-5}
if [[ $PARTS == *bench* ]]; then
  timeout -k 10 900 python bench.py $BARGS --steps $STEPS --warmup 1 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { tail -5 $O/${TAG}_bench.err; exit 1; }
  tail -1 $O/${TAG}_bench.json | cut -c1-400
fi
if [[ $PARTS == *stats* ]]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 bench.py $BARGS --steps $STEPS --warmup 1 --cpu-spp 0 --no-calib --no-throughput-line > $O/${TAG}_stats.log 2>&1 || { tail -5 $O/${TAG}_stats.log; exit 1; }
  cp $(find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv && head -4 $O/${TAG}_kernel_stats.csv
  rm -rf $O/${TAG}_stats
fi
if [[ $PARTS == *hbm* ]]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/${TAG}_pmc_$c -- ./mort_amd/bin/mort $MARGS > $O/${TAG}_pmc_$c.log 2>&1 || { tail -5 $O/${TAG}_pmc_$c.log; exit 1; }
  done
  python3 scripts/pmc_summary.py $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE > $O/${TAG}_pmc_hbm_raw.csv
  python3 scripts/pmc_hbm_json.py $O/${TAG}_pmc_hbm_raw.csv $SCENE $WIDTH $SPP $MODE "$MARGS" > $O/${TAG}_pmc_hbm.json
  cat $O/${TAG}_pmc_hbm.json
  rm -rf $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE
fi
if [[ $PARTS == *sq* ]]; then
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/${TAG}_pmc_sq1 -- ./mort_amd/bin/mort $MARGS > $O/${TAG}_pmc_sq1.log 2>&1 || { tail -5 $O/${TAG}_pmc_sq1.log; exit 1; }
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SMEM SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_ACTIVE_INST_ANY --output-format csv -d $O/${TAG}_pmc_sq2 -- ./mort_amd/bin/mort $MARGS > $O/${TAG}_pmc_sq2.log 2>&1 || { tail -5 $O/${TAG}_pmc_sq2.log; exit 1; }
  python3 scripts/pmc_summary.py $O/${TAG}_pmc_sq1 $O/${TAG}_pmc_sq2 > $O/${TAG}_pmc_sq_summary.csv
  grep -E "SQ_INSTS_VALU|SQ_THREAD_CYCLES_VALU|SQ_ACTIVE_INST_VALU|SQ_BUSY_CYCLES" $O/${TAG}_pmc_sq_summary.csv
  rm -rf $O/${TAG}_pmc_sq1 $O/${TAG}_pmc_sq2
fi
