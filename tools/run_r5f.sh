#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_pipeline.py tests/test_gpu_render.py tests/test_gpu_zoom.py -x -q 2>&1 | tail -8 | tee gpurun_out/r5f_pytest.log
