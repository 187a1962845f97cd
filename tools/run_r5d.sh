#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_pipeline.py tests/test_gpu_net.py -x -q 2>&1 | tail -5
timeout 300 python tools/bench_wino.py 32 2>&1 | tee gpurun_out/r5d_bench_wino_b32.log
timeout 300 python tools/bench_wino.py 4 2>&1 | tee gpurun_out/r5d_bench_wino_b4.log
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value',d['value'],'ms/step',d['ms_per_step'],'conv ms',d['roofline']['ms_per_launch_group'],'frac_exec',d['roofline'].get('frac_executed'))
print('parity',d['parity']['pose_max_rel'],d['parity']['se3_max_rel'],d['parity']['within_bar'])
for k,v in d['other_configs'].items(): print(k, v['value'])
" | tee gpurun_out/r5d_bench.log
