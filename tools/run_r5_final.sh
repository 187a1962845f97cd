#!/bin/bash
# round-5 closing evidence: the default bench line as the driver runs it, then the rocprofv3 summaries (tools/run_r5_profiles.sh)
cd /root/repo; mkdir -p gpurun_out
( time timeout 600 python bench.py 2> gpurun_out/r05_bench_default.err | grep "^{" | tail -1 > gpurun_out/r05_bench_default_n1.json.log ) 2>&1 | tail -3
bash tools/run_r5_profiles.sh
