cd /root/repo
python -m pytest tests/test_gpu_wino.py tests/test_gpu_baseline_configs.py -x -q 2>&1 | tail -4 > gpurun_out/r06_b_tests.log
AB_BATCH=32 bash tools/run_ab.sh > /dev/null 2>&1
AB_BATCH=4 WINO_LAYERS=conv3_1,conv4_1,conv5_1,conv6_1,conv2,conv3 bash tools/run_ab.sh > /dev/null 2>&1
B="python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 --verify 1 --full"
for cfg in "--batch 4 --graph off --opt wino_fin=0" "--batch 4 --graph off" "--batch 4" "--batch 8 --graph off --opt wino_fin=0" "--batch 8" "--batch 4 --heads --graph off --opt wino_fin=0" "--batch 4 --heads" "--batch 32 --steps 10 --opt wino_fin=0" "--batch 32 --steps 10"; do
  echo "== $cfg"; timeout 300 $B $cfg 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); p=j.get('parity') or {}
print('%.1f it/s  %.3f ms/step  enc %.3f ms  frac %.3f  launch=%s  pose %.1e ok=%s' % (j['value'], j['ms_per_step'], j['roofline']['ms_per_launch_group'], j['roofline']['frac'], j['config']['encoder_launch'], p.get('pose_max_rel',-1), p.get('within_bar')))"
done > gpurun_out/r06_b_bench.log 2>&1
cat gpurun_out/r06_b_tests.log gpurun_out/ab_b32.log gpurun_out/ab_b4.log gpurun_out/r06_b_bench.log
