#!/bin/bash
# round 5, session a: the shared-transform Winograd kernel — parity tests, then per-layer timings against the round-4 kernel
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -15 > gpurun_out/r5a_pytest_wino.log
cat gpurun_out/r5a_pytest_wino.log
timeout 300 python tools/bench_wino.py 32 2>&1 | tee gpurun_out/r5a_bench_wino_b32.log
timeout 200 python tools/bench_wino.py 4 2>&1 | tee gpurun_out/r5a_bench_wino_b4.log
