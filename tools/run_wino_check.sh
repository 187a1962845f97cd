cd /root/repo
timeout 300 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -5
timeout 200 python tools/bench_wino.py 32
