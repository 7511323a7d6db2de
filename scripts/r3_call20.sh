#!/bin/bash
# round 3, call 21: four-wide nodes at 144 B, leaf records with spheres by value (80 B) vs the binary build; LDS conflict counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q > $O/r3v_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r3v_pytest.log
[ $rc -eq 0 ] || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:70], d['kernel']['lds_bytes'])"; }
for L in build/variants/bin2/lib/libmort_hip.so mort_amd/lib/libmort_hip.so build/variants/bin2/lib/libmort_hip.so mort_amd/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## $L"
  b || exit 1; b --mode throughput; b --mode wave
done
for L in build/variants/bin2/lib/libmort_hip.so mort_amd/lib/libmort_hip.so; do export MORT_HIP_LIB=$L; echo "######## partition $L"
  timeout -k 10 300 python scripts/time_partition.py 1 1200 500 mega | python -c "import sys,json; d=json.load(sys.stdin); print({k:[round(t['ms'],1) for t in v] for k,v in d['ranks'].items()})" || exit 1
done
unset MORT_HIP_LIB
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/r3v_pmc_lds -- ./mort_amd/bin/mort 1 --width 1200 --spp 500 --mode mega > $O/r3v_pmc_lds.log 2>&1 || { tail -5 $O/r3v_pmc_lds.log; exit 1; }
python3 scripts/pmc_summary.py $O/r3v_pmc_lds > $O/r3v_pmc_lds_summary.csv; grep -v "true false\|iota\|seed\|tile" $O/r3v_pmc_lds_summary.csv; rm -rf $O/r3v_pmc_lds
