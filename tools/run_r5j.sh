#!/bin/bash
# same-box A/B of the whole bench line: stream-K of the Winograd layers' last round on / off, interleaved, at several batch sizes
cd /root/repo; mkdir -p gpurun_out
for B in ${AB_BATCHES:-16 8 4}; do for rep in 1 2; do for v in 1 0; do
  echo "== B=$B wino_streamk=$v (pass $rep)"
  timeout 300 python bench.py --batch $B --no-cpu-baseline --no-other-configs --verify 0 --opt wino_streamk=$v 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['roofline']['ms_per_launch_group'])"
done; done; done | tee gpurun_out/r5j.log
