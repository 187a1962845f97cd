set -x
cd /root/repo
python -m pytest tests/test_gpu_backward.py -m gpu -q -x 2>&1 | tail -5
python tools/bench_train.py 4 heads 2>&1 | tail -2
python tools/bench_train.py 4 2>&1 | tail -1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_train -- python tools/bench_train.py 4 heads > gpurun_out/r3_train.log 2>&1
f=$(find gpurun_out/r3_train -name "*kernel_stats.csv" | head -1)
head -30 $f > gpurun_out/r3_train_kernel_stats.csv
rm -rf gpurun_out/r3_train
