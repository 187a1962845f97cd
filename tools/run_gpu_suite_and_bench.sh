#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r5c_pytest_gpu.log
( time timeout 600 python bench.py 2>&1 | tail -3 ) 2>&1 | tee gpurun_out/r5c_bench.log
