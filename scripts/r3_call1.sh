#!/bin/bash
# round 3, first GPU call: the GPU suite on the refactored host code, the roofline calibration, the headline bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/r3a_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r3a_pytest.log
timeout -k 10 120 python scripts/calib.py $O/r3a_calib.json 2> $O/r3a_calib.log > /dev/null; echo "calib rc=$?"; cat $O/r3a_calib.log
timeout -k 10 300 python bench.py > $O/r3a_bench.json 2> $O/r3a_bench.err; echo "bench rc=$?"; tail -1 $O/r3a_bench.json | cut -c1-600
