#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_net.py -x -q -k "few_output" 2>&1 | tail -3 | tee gpurun_out/r5s_test.log
for rep in 1 2; do for v in 1 0; do
  echo "== heads B=32 conv_fewout_quad=$v (pass $rep)"
  timeout 300 python bench.py --heads --no-cpu-baseline --no-other-configs --verify 0 --opt conv_fewout_quad=$v 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"
done; done | tee gpurun_out/r5s.log
bash tools/run_r5t.sh 2>&1 | grep -E "fewout|splitk_reduce_kernel|total"
