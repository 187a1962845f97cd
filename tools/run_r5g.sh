#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_pipeline.py tests/test_gpu_baseline_configs.py -x -q 2>&1 | tail -4
timeout 300 python bench.py --heads --no-cpu-baseline --no-other-configs --verify 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('heads B32', d['value'], d['ms_per_step'], d['parity'].get('within_bar'), d['parity'].get('flow_max_rel'))"
timeout 300 python bench.py --heads --batch 4 --no-cpu-baseline --no-other-configs --verify 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('heads B4', d['value'], d['ms_per_step'])"
