#!/bin/bash
# same-box A/B of the whole bench line with a context option on / off (AB_OPT, default wino_streamk), interleaved, at several batch sizes (AB_BATCHES)
cd /root/repo; mkdir -p gpurun_out
for B in ${AB_BATCHES:-16 8 4}; do for rep in 1 2; do for v in 1 0; do
  echo "== B=$B ${AB_OPT:-wino_streamk}=$v (pass $rep)"
  timeout 300 python bench.py --batch $B --no-cpu-baseline --no-other-configs --verify 0 --opt ${AB_OPT:-wino_streamk}=$v 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['roofline']['ms_per_launch_group'])"
done; done; done | tee gpurun_out/ab_bench_option.log
