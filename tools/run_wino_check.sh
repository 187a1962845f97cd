cd /root/repo
timeout 200 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -5
timeout 120 python tools/bench_wino.py 32
timeout 120 python tools/bench_wino.py 4
