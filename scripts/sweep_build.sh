#!/bin/bash
# build tuning variants of libmort_hip.so: scripts/sweep_build.sh "name:-DFLAGS" ...
set -e
mkdir -p build/variants
build() {
  name="${1%%:*}"; flags="${1#*:}"
  mkdir -p build/variants/$name
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-gpu-rdc \
     -Wno-unused-value -Wno-unused-result -Iinclude $flags -shared -o build/variants/$name/libmort_hip.so mort_amd/csrc/hip/mort_hip.hip \
     -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "mega_bvh" | grep -E "VGPRs:|Scratch|Occupancy" | tr '\n' ' ' | sed "s/^/$name: /"; echo
}
for v in "$@"; do build "$v" & 
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.5; done
done
wait
