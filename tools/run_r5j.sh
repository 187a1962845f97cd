#!/bin/bash
# same-box A/B of the whole bench line: persistent Winograd grid on / off, interleaved
cd /root/repo; mkdir -p gpurun_out
for rep in 1 2 3; do for v in 1 0; do
  echo "== wino_persistent=$v (pass $rep)"
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --verify 0 --opt wino_persistent=$v 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['roofline']['ms_per_launch_group'])"
done; done | tee gpurun_out/r5j.log
for v in 1 0; do echo "== B=4 persistent=$v"; timeout 300 python bench.py --batch 4 --no-cpu-baseline --no-other-configs --verify 0 --opt wino_persistent=$v 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['roofline']['ms_per_launch_group'])"; done | tee -a gpurun_out/r5j.log
