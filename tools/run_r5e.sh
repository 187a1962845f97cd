#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -5
timeout 300 python tools/bench_wino.py 32 2>&1 | tee gpurun_out/r5e_bench_wino_b32.log
timeout 300 python tools/bench_wino.py 8 2>&1 | tee gpurun_out/r5e_bench_wino_b8.log
