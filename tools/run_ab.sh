#!/bin/bash
# A/B of library builds in ONE box (boxes differ by a few per cent): variants/lib_<name>.so, interleaved, twice
# AB_BATCH=32 WINO_LAYERS=conv3_1,conv4_1,conv2,conv3 bash tools/run_ab.sh
cd /root/repo; mkdir -p gpurun_out
for rep in 1 2; do for v in $(ls variants | sed 's/lib_//; s/.so//'); do
  echo "== $v (pass $rep)"
  DEEPIM_LIB=variants/lib_$v.so WINO_LAYERS=${WINO_LAYERS:-conv3_1,conv4_1,conv2,conv3} timeout 200 python tools/bench_wino.py ${AB_BATCH:-32} 2>&1 | cut -c1-75
done; done | tee gpurun_out/ab_b${AB_BATCH:-32}.log
