cd /root/repo; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3_tt -- python tools/bench_train.py 4 heads > gpurun_out/r3_tt.log 2>&1
f=$(find gpurun_out/r3_tt -name "*kernel_trace.csv" | head -1)
python tools/trace_train.py $f > gpurun_out/r3_train_iteration_trace.txt
rm -rf gpurun_out/r3_tt
wc -l gpurun_out/r3_train_iteration_trace.txt
