#!/bin/bash
# flag_variants.sh FILE.hip NAME=FLAGS... : libraries under build/fv/<NAME>/ whose FILE.o is compiled with the extra flags (experiments)
SRC=$1; shift
base=$(basename $SRC .hip)
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fno-gpu-rdc -w -Iinclude -Imort_amd/csrc/hip"
others=$(ls build/hip/*.o | grep -v "/$base.o")
rm -rf build/fv; mkdir -p build/fv
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  mkdir -p build/fv/$name
  ( if hipcc $F $flags -c -o build/fv/$name/$base.o $SRC 2> build/fv/$name/err.txt; then
      hipcc --offload-arch=gfx950 -fno-gpu-rdc -shared -o build/fv/$name/libmort_hip.so build/fv/$name/$base.o $others -lpthread -ldl; rm build/fv/$name/$base.o
    else echo "$name: does not compile"; tail -2 build/fv/$name/err.txt; rm -rf build/fv/$name; fi ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
ls build/fv
