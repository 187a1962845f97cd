cd /root/repo
for v in default s2d256; do
  L=""; [ $v != default ] && export DEEPIM_LIB=/root/repo/variants/lib_$v.so
  for B in 4 8 32; do
    timeout 150 python bench.py --batch $B --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 --verify 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v B=%2d  %8.1f it/s  wino=%s' % ($B, j['value'], ','.join(j['roofline']['winograd_layers'])))"
  done
done
