#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
( time timeout 1500 python -m pytest ${PYTEST_FILES:-tests/test_gpu_backward.py tests/test_gpu_net.py tests/test_gpu_pipeline.py tests/test_gpu_operators.py} -q 2>&1 | tail -12 ) 2>&1 | tee gpurun_out/r5p_pytest_gpu.log
