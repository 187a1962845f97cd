#!/bin/bash
# Dev helper: libdeepim_hip variants with different -D flags for one source (SRC=conv_f16 by default) into variants/lib_<name>.so
# usage: [SRC=wino] tools/build_variants_f16.sh name1:"-DA=1" name2:"..."    (run with DEEPIM_LIB=variants/lib_<name>.so)
set -e
cd "$(dirname "$0")/../mx_deepim_amd/csrc"
make -s
mkdir -p ../../variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-value -Wno-unused-result -Wno-uninitialized"
SRC="${SRC:-conv_f16}"
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  /opt/rocm/bin/hipcc $FLAGS $defs -c $SRC.hip -o /tmp/${SRC}_$name.o
  objs=$(ls *.o | grep -v "^$SRC.o\$" | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so /tmp/${SRC}_$name.o $objs -ldl
  echo "built variants/lib_$name.so ($defs)"
done
