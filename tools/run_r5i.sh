#!/bin/bash
cd /root/repo
for v in w8trace w8trace_af; do echo "== $v"; DEEPIM_LIB=variants/lib_$v.so timeout 200 python tools/wino8_trace.py 32 2>&1 | head -18; done | tee gpurun_out/r5i_w8trace.log
