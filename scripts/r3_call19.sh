#!/bin/bash
# round 3, call 20: LDS bank-conflict counters of the two state-machine kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_LDS[A-Z_]*\|SQ_[A-Z_]*LDS[A-Z_]*" | sort -u | tr '\n' ' '; echo
run() { tag=$1; shift
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/${tag}_pmc_lds -- ./mort_amd/bin/mort "$@" > $O/${tag}_pmc_lds.log 2>&1 || { tail -5 $O/${tag}_pmc_lds.log; return 1; }
  python3 scripts/pmc_summary.py $O/${tag}_pmc_lds > $O/${tag}_pmc_lds_summary.csv; grep -v "true false\|iota\|seed\|tile" $O/${tag}_pmc_lds_summary.csv; rm -rf $O/${tag}_pmc_lds; }
run r3u_s1 1 --width 1200 --spp 500 --mode mega || exit 1
run r3u_s8 8 --width 800 --spp 100 --mode mega || exit 1
