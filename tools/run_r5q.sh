#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for rep in 1 2; do for sk in 1 2; do for B in 32 8 4; do
  echo "== streamk=$sk B=$B (pass $rep)"
  L=conv3_1,conv4_1,conv5_1,conv2,conv3; [ $B = 32 ] && L=conv5_1
  WINO_LAYERS=$L WINO_STREAMK=$sk timeout 200 python tools/bench_wino.py $B 2>&1 | cut -c1-75
done; done; done | tee gpurun_out/r5q.log
