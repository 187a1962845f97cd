#!/bin/bash
# Dev helper: libdeepim_hip variants with different -D flags for wino.hip into variants/lib_<name>.so
# usage: tools/build_variants_wino.sh name1:"-DW8_ABL=1" name2:"..."
set -e
cd "$(dirname "$0")/../mx_deepim_amd/csrc"
make -s
mkdir -p ../../variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-value -Wno-unused-result"
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  /opt/rocm/bin/hipcc $FLAGS $defs -c wino.hip -o /tmp/wino_$name.o
  objs=$(ls *.o | grep -v '^wino.o$' | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so /tmp/wino_$name.o $objs -ldl
  echo "built variants/lib_$name.so ($defs)"
done
