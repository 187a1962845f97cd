cd /root/repo
for v in wtrace wtrnone wtrnoAB; do echo "== $v"; DEEPIM_LIB=variants/lib_$v.so timeout 100 python tools/wino_trace.py 32 2>&1 | tail -10; done
