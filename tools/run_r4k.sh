cd /root/repo
timeout 600 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -15
timeout 300 python tools/bench_wino.py 32
timeout 300 python tools/bench_wino.py 4
