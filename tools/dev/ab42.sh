# ablations / variants of the F(4,3)xF(2,3) kernel in ONE box: variants/lib_<name>.so, conv3_1 + conv4_1 at B = 32 (F(4,3)xF(2,3) line only)
cd /root/repo
for rep in 1 2; do for v in $(ls variants | sed 's/lib_//; s/.so//'); do
  echo "== $v (pass $rep)"
  DEEPIM_LIB=variants/lib_$v.so WINO_LAYERS=${WINO_LAYERS:-conv3_1,conv4_1} timeout 200 python tools/bench_wino.py ${AB_BATCH:-32} 2>&1 | grep "F(4,3)\|winograd" | cut -c1-110
done; done
