#!/bin/bash
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out
for cfg in "0 r9" "0 r9375" "0 r10" "1 r9375" "0 c41" "1 c41"; do
  set -- $cfg; sk=$1; L=$2
  rm -rf /tmp/prof_x
  WINO_STREAMK=$sk WINO_LAYERS=$L WINO_CUSTOM="r9,256,64,72,256;r9375,256,60,80,256;r10,256,64,80,256;c41,512,30,40,512" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python /root/repo/tools/bench_wino.py 32 > /tmp/out_x.log 2>&1
  f=$(find /tmp/prof_x -name "*kernel_stats.csv" | head -1)
  echo "== streamk=$sk $L"; grep -E "conv_wino8|fixup" "$f" | cut -d, -f1-4,6-8
done 2>&1 | tee /root/repo/gpurun_out/r5n.log
