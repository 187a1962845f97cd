#!/bin/bash
# round 4, GPU session A: ping-pong fp16 kernel (parity first, then per-layer A/B), the whole GPU suite, the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 420 python -m pytest tests/test_gpu_fp16.py -q -k pingpong 2>&1 | tail -25 ) > gpurun_out/r4a_pp.log
( timeout 240 python tools/bench_layers_f16.py 32 8 0,16 2>&1 | tail -20 ) > gpurun_out/r4a_layers32.log
( timeout 120 python tools/bench_layers_f16.py 8 8 0,16 2>&1 | tail -20 ) > gpurun_out/r4a_layers8.log
( timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fp16.py::test_conv_f16_pingpong_kernel 2>&1 | tail -40 ) > gpurun_out/r4a_tests.log
( timeout 200 python tools/bench_train.py 4 heads step4 2>&1 | tail -5 ) > gpurun_out/r4a_train_step4.log
timeout 500 python bench.py > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err
tail -3 gpurun_out/r4a_pp.log; cat gpurun_out/r4a_layers32.log; tail -5 gpurun_out/r4a_tests.log; cat gpurun_out/r4a_train_step4.log; head -c 600 gpurun_out/r4a_bench.json
