#!/bin/bash
# round 3, call 10: wavefront traversal kernel without private memory (98 VGPRs, 1024-thread workgroups on big images); thresholds as shares of live lanes
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 500 python -m pytest tests/test_gpu_gen.py -x -q -k "wavefront or config5 or random_worlds" > gpurun_out/r3j_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3j_pytest.log
[ $rc -eq 0 ] || exit 1
b() { timeout -k 10 300 python bench.py --no-calib --cpu-spp 0 --no-throughput-line --steps 2 --warmup 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], d['config']['workload'][:70])"; }
echo "== wave mode, config 5 geometry 4096x4096x4: base / new (1024) / new 512"
MORT_HIP_LIB=build/variants/base/lib/libmort_hip.so b --scene 8 --width 4096 --aspect 1 --spp 4 --mode wave
b --scene 8 --width 4096 --aspect 1 --spp 4 --mode wave
MORT_WAVE_TRAV_BLOCK=512 b --scene 8 --width 4096 --aspect 1 --spp 4 --mode wave
echo "== megakernel same frame"; MORT_GEN_BLOCK_SIZE=768 b --scene 8 --width 4096 --aspect 1 --spp 4
echo "== wave mode s8 800x800x100: base / new"; MORT_HIP_LIB=build/variants/base/lib/libmort_hip.so b --scene 8 --width 800 --spp 100 --mode wave; b --scene 8 --width 800 --spp 100 --mode wave
echo "== wave mode Cornell 800x800x100: base / new"; MORT_HIP_LIB=build/variants/base/lib/libmort_hip.so b --scene 6 --width 800 --spp 100 --mode wave; b --scene 6 --width 800 --spp 100 --mode wave
t() { timeout -k 10 300 python scripts/gen_profile.py $1 $2 $3 4 2>&1 | grep "^frame [23]" | cut -c1-100; }
for dm in 0 3; do echo "== s8 800x800x100 MORT_GEN_DRAIN=$dm block 768"; MORT_GEN_DRAIN=$dm MORT_GEN_BLOCK_SIZE=768 t 8 800 100; done
for dm in 0 3; do echo "== s8 1920x1080x49 N=8 MORT_GEN_DRAIN=$dm"; MORT_GEN_DRAIN=$dm MORT_GEN_BLOCK_SIZE=768 timeout -k 10 100 python scripts/tp_one.py 8 1920 49 8 1.7777778 2>&1 | tail -1 | cut -c1-160; done
for dm in 1 3; do for n in 1 4 8; do echo "== headline N=$n MORT_BVH_DRAIN=$dm"; MORT_BVH_DRAIN=$dm timeout -k 10 100 python scripts/tp_one.py 1 1200 500 $n 2>&1 | tail -1 | cut -c1-160; done; done
