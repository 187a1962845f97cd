#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_fullsize_properties.py tests/test_gpu_net.py tests/test_gpu_operators.py tests/test_gpu_ops_golden.py tests/test_gpu_pipeline.py tests/test_gpu_render.py tests/test_gpu_se3_heads.py tests/test_gpu_se3_surface.py tests/test_gpu_wino.py tests/test_gpu_x3.py tests/test_gpu_zoom.py tests/test_gpu_zoom_golden.py -q 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/r5p_pytest_gpu.log
