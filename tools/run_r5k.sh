#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -3 | tee gpurun_out/r5k_test.log
WINO_LAYERS=conv3_1,conv4_1,conv5_1,conv2,conv3 bash tools/run_ab.sh
