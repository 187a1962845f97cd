#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -5
timeout 300 python tools/bench_wino.py 32 2>&1 | tee gpurun_out/r5b_bench_wino_b32.log
for v in $(ls variants | sed 's/lib_//; s/.so//'); do
  echo "== $v"
  DEEPIM_LIB=variants/lib_$v.so WINO_LAYERS=conv3_1,conv2,conv3 timeout 200 python tools/bench_wino.py 32 2>&1 | cut -c1-75
done | tee gpurun_out/r5b_ablation.log
