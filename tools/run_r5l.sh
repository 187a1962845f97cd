#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for v in w8trace1 w8trace2; do echo "== $v"; DEEPIM_LIB=variants/lib_$v.so timeout 200 python tools/wino8_trace.py 32; done 2>&1 | tee gpurun_out/r5l.log
