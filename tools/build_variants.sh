#!/bin/bash
# Dev helper: build libdeepim_hip variants with different -D flags for conv.hip into gpurun_variants/<name>.so
# usage: tools/build_variants.sh name1:"-DA=1 -DB=2" name2:"..."
set -e
cd "$(dirname "$0")/../mx_deepim_amd/csrc"
make -s
mkdir -p ../../variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-value -Wno-unused-result"
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  /opt/rocm/bin/hipcc $FLAGS $defs -c conv.hip -o /tmp/conv_$name.o
  objs=$(ls *.o | grep -v '^conv.o$' | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so /tmp/conv_$name.o $objs
  echo "built variants/lib_$name.so ($defs)"
done
