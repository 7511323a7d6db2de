#!/bin/bash
# round 3, call 25: 1024-thread BVH megakernel selected for full frames (15 spilled registers, shade step only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:40], d['kernel']['lds_bytes'], d['kernel'].get('vgprs'))"; }
echo "######## main"; b || exit 1; b --mode throughput; b --scene 10 --width 1200 --spp 100
echo "######## 768"; MORT_FAST_BLOCK_SIZE=768 b
for th in 48,16,24 48,20,16 44,16,16 48,12,16 52,16,16; do echo "== thresholds $th"; MORT_THRESHOLDS=$th b; done
timeout -k 10 300 python scripts/time_partition.py 1 1200 500 mega | python -c "import sys,json; d=json.load(sys.stdin); print({k:[(round(t['ms'],1), t['kernel'][16:20]) for t in v] for k,v in d['ranks'].items()})" || exit 1
echo "== partition, 1024 forced"; MORT_FAST_BLOCK_SIZE=1024 timeout -k 10 300 python scripts/time_partition.py 1 1200 500 mega | python -c "import sys,json; d=json.load(sys.stdin); print({k:[(round(t['ms'],1), t['kernel'][16:20]) for t in v] for k,v in d['ranks'].items()})"
MORT_FAST_BLOCK_SIZE=1024 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q > $O/r3B_pytest1024.log 2>&1; rc=$?; echo "pytest(1024 forced) rc=$rc"; tail -3 $O/r3B_pytest1024.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r3B_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r3B_pytest.log
