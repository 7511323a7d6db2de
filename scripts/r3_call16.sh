#!/bin/bash
# round 3, call 16: box-step loop hands control back only when another state has its batch together
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_gen.py tests/test_gpu_parity.py -x -q -k "not config5 and not wavefront" > gpurun_out/r3p_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3p_pytest.log
[ $rc -eq 0 ] || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:70], d['kernel']['lds_bytes'])"; }
b; b
b --scene 8 --width 800 --spp 100; MORT_GEN_BLOCK_SIZE=768 b --scene 8 --width 800 --spp 100
b --scene 8 --width 1920 --aspect 1.7777778 --spp 49
b --scene 8 --width 4096 --aspect 1 --spp 4
b --scene 8 --width 800 --spp 1000 --steps 2
for n in 2 8; do MORT_GEN_BLOCK_SIZE=768 timeout -k 10 100 python scripts/tp_one.py 8 1920 49 $n 1.7777778 2>&1 | tail -1 | cut -c1-160; done
for n in 2 4 8; do timeout -k 10 100 python scripts/tp_one.py 1 1200 500 $n 2>&1 | tail -1 | cut -c1-160; done
