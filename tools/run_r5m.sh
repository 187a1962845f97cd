#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -5 | tee gpurun_out/r5m_test.log
for rep in 1 2; do for sk in 0 1; do for B in 32 ${AB_BATCHES:-}; do
  echo "== streamk=$sk B=$B (pass $rep)"
  WINO_STREAMK=$sk timeout 200 python tools/bench_wino.py $B 2>&1 | cut -c1-75
done; done; done | tee gpurun_out/r5m_ab.log
