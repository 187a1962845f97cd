#!/bin/bash
# round-5 closing evidence: the GPU suite, the default bench line as the driver runs it, smoke(), then the rocprofv3 summaries (tools/run_r5_profiles.sh)
cd /root/repo; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r05_pytest_gpu.log
( time timeout 600 python bench.py 2> gpurun_out/r05_bench_default.err | grep "^{" | tail -1 > gpurun_out/r05_bench_default_n1.json.log ) 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r05_smoke.log
bash tools/run_r5_profiles.sh
