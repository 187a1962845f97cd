cd /root/repo
timeout 300 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -12
timeout 300 python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -5
timeout 200 python bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B32', j['value'], j['roofline']['achieved'], j['roofline'].get('executed'), j['roofline'].get('winograd_layers'), {k: j['parity'][k] for k in ('pose_max_rel','se3_max_rel','within_bar')})"
timeout 200 python bench.py --batch 4 --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B4', j['value'], j['roofline']['achieved'], j['roofline'].get('winograd_layers'), {k: j['parity'][k] for k in ('pose_max_rel','se3_max_rel','within_bar')})"
