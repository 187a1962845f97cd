#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for a in "--graph on" "--batch 2" "--batch 1" "--heads --batch 8" "--no-winograd" "--iters 2 --batch 24"; do
  echo "== bench.py $a"
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 4 --warmup 1 $a 2>&1 | grep -E '^\{|Error|error' | python -c "
import sys,json
for l in sys.stdin:
    try:
        d=json.loads(l); print(round(d['value'],1), d.get('parity',{}).get('within_bar'), d.get('parity',{}).get('pose_max_rel'))
    except Exception: print(l[:200])"
done 2>&1 | tee gpurun_out/r5u.log
