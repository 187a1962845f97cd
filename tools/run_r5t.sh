#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/prof_h; ( cd /tmp; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_h -- python /root/repo/bench.py --no-cpu-baseline --no-other-configs --verify 0 --batch 32 --heads --steps 4 --warmup 2 > /tmp/heads.log 2>&1 )
hr=$(find /tmp/prof_h -name "*kernel_trace.csv" | head -1)
python tools/trace_iteration.py $hr 1 > gpurun_out/r5t_heads_trace.txt
tail -32 gpurun_out/r5t_heads_trace.txt
