cd /root/repo; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof -- python tools/bench_train.py 4 heads > gpurun_out/r3_train_prof.log 2>&1
st=$(find gpurun_out/r3_prof -name "*kernel_stats.csv" | head -1); tr=$(find gpurun_out/r3_prof -name "*kernel_trace.csv" | head -1)
head -60 $st > gpurun_out/r03_train_iteration_kernel_stats.csv
python tools/trace_train.py $tr > gpurun_out/r03_train_iteration_trace.txt
rm -rf gpurun_out/r3_prof
python tools/bench_train.py 4 heads | tail -1; python tools/bench_train.py 4 | tail -1
