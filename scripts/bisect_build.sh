#!/bin/bash
# bisect_build.sh FILE.hip "EXTRA_FLAGS" N1 N2 ... : variant libraries under build/bis/<N>/ whose FILE.o is compiled with
# -mllvm -opt-bisect-limit=N (LLVM's pass bisection), everything else from build/hip.  Diagnostic tool (dev_shade_call.h's story).
set -e
SRC=$1; EXTRA=$2; shift 2
base=$(basename $SRC .hip)
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-gpu-rdc -w -Iinclude -Imort_amd/csrc/hip $EXTRA"
others=$(ls build/hip/*.o | grep -v "/$base.o")
rm -rf build/bis; mkdir -p build/bis
for N in "$@"; do
  mkdir -p build/bis/$N
  if ! hipcc $F -mllvm -opt-bisect-limit=$N -c -o build/bis/$N/$base.o $SRC 2> build/bis/$N/bisect.txt; then echo "limit $N: does not compile"; rm -rf build/bis/$N; continue; fi
  hipcc --offload-arch=gfx950 -fno-gpu-rdc -shared -o build/bis/$N/libmort_hip.so build/bis/$N/$base.o $others -lpthread -ldl
  rm build/bis/$N/$base.o; grep "BISECT: running" build/bis/$N/bisect.txt | grep -v "NOT running" | tail -1 > build/bis/$N/last_pass.txt; rm build/bis/$N/bisect.txt
done
ls build/bis
