#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for rep in 1 2; do for sk in 0 1; do
  echo "== streamk=$sk (pass $rep): 9.0 / 9.375 / 10.0 rounds of wide blocks"
  WINO_STREAMK=$sk WINO_LAYERS=r9,r9375,r10 WINO_CUSTOM="r9,256,64,72,256;r9375,256,60,80,256;r10,256,64,80,256" timeout 200 python tools/bench_wino.py 32 2>&1 | cut -c1-140
done; done | tee gpurun_out/r5o.log
