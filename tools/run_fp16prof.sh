set -x
cd /root/repo
python -m pytest tests/test_gpu_fp16.py tests/test_gpu_baseline_configs.py -m gpu -q -x 2>&1 | tail -4
export TMPDIR=/tmp
COMMON="--steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --verify 0"
for B in 8 32; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3_f16d_b$B -- python bench.py --fp16 --depth --batch $B $COMMON --graph off > gpurun_out/r3_f16d_b$B.log 2>&1
  f=$(find gpurun_out/r3_f16d_b$B -name "*kernel_trace.csv" | head -1)
  python tools/trace_iteration.py $f 1 > gpurun_out/r3_f16d_b$B.iter.txt
  rm -rf gpurun_out/r3_f16d_b$B
done
for g in off on; do python bench.py --fp16 --depth --batch 8 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --verify 0 --graph $g 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('graph $g B8', j['value'], j['roofline']['achieved'], j['roofline']['ms_per_launch_group'])"; done
python bench.py --fp16 --depth --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --verify 0 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('B32', j['value'], j['roofline']['achieved'], j['roofline']['ms_per_launch_group'])"
python bench.py --batch 4 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --verify 0 --graph off 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('fp32 B4 graph off', j['value'], j['roofline']['achieved'])"
python bench.py --batch 4 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --verify 0 --graph on 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('fp32 B4 graph on', j['value'], j['roofline']['achieved'])"
