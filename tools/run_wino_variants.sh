cd /root/repo
for v in $VARIANTS; do echo "== $v"; DEEPIM_LIB=variants/lib_$v.so timeout 300 python tools/bench_wino.py 32 2>&1 | head -2; done
