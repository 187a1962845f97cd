# short closed-loop runs over per-GPU batch sizes (the shards of a strong-scaling run): value, Winograd layers bound, parity
cd /root/repo
for B in ${BATCHES:-1 2 3 4 8 16}; do
  timeout 150 python bench.py --batch $B --no-cpu-baseline --no-other-configs --steps 8 --warmup 2 --verify 1 --full 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=j.get('parity') or {}
print('B=%2d  %8.1f it/s  conv %.1f TF alg / %.1f exec  wino=%s  pose %.1e se3 %.1e ok=%s' % ($B, j['value'], j['roofline']['algorithmic_tflops'], j['roofline']['achieved'], ','.join(j['roofline']['winograd_layers']) or '-', p.get('pose_max_rel', -1), p.get('se3_max_rel', -1), p.get('within_bar')))"
done
