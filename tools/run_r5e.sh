#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -3
for b in 32 8 4; do timeout 300 python tools/bench_wino.py $b 2>&1 | cut -c1-60,96-175 | tee gpurun_out/r5e_bench_wino_b$b.log; done
