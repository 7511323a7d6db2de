#!/bin/bash
# round 3, call 7: per-wave lifetimes (device log, no printf): how much of each frame is tail
cd "$GRAFT_REPO_ROOT" || exit 1
export MORT_HIP_LIB=build/variants/prof/lib/libmort_hip.so
echo "== headline"; timeout -k 10 200 python scripts/wave_lines.py 1 1200 500 2>&1 | tail -7 | cut -c1-330
echo "== headline, no tile order"; MORT_NO_TILE_ORDER=1 timeout -k 10 200 python scripts/wave_lines.py 1 1200 500 2>&1 | tail -7 | cut -c1-330
echo "== final scene 800x800x100, block 768"; MORT_GEN_BLOCK_SIZE=768 MORT_GEN_PRIO_LANES=0 timeout -k 10 200 python scripts/wave_lines.py 8 800 100 2>&1 | tail -7 | cut -c1-330
echo "== final scene 800x800x100, block 768, prio 2"; MORT_GEN_BLOCK_SIZE=768 MORT_GEN_PRIO_LANES=2 timeout -k 10 200 python scripts/wave_lines.py 8 800 100 2>&1 | tail -7 | cut -c1-330
echo "== final scene 1920x1080x49, block 768"; MORT_GEN_BLOCK_SIZE=768 MORT_GEN_PRIO_LANES=0 timeout -k 10 200 python scripts/wave_lines.py 8 1920 49 1.7777778 2>&1 | tail -7 | cut -c1-330
echo "== final scene 4096x4096x4, block 768"; MORT_GEN_BLOCK_SIZE=768 MORT_GEN_PRIO_LANES=0 timeout -k 10 200 python scripts/wave_lines.py 8 4096 4 1.0 2>&1 | tail -7 | cut -c1-330
echo "== Cornell on mega_gen 800x800x100"; MORT_GEN_MIN_PRIMS=0 MORT_GEN_BLOCK_SIZE=768 MORT_GEN_PRIO_LANES=0 timeout -k 10 200 python scripts/wave_lines.py 6 800 100 2>&1 | tail -7 | cut -c1-330
