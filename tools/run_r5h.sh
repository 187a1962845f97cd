#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
for b in 32 16 8 4; do
timeout 300 python bench.py --batch $b --no-cpu-baseline --no-other-configs --verify 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B', d['config']['pairs_per_gpu'], 'it/s %.0f'%d['value'], 'conv ms %.3f'%d['roofline']['ms_per_launch_group'], 'frac %.3f'%d['roofline']['frac'], d['parity'].get('within_bar'), d['parity'].get('pose_max_rel'))"
done | tee gpurun_out/r5h_bench.log
