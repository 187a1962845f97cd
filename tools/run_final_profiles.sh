# round-3 closing run on the MI355X box: full GPU suite, the default bench line, and the rocprofv3 summaries committed under profiles/
cd /root/repo; export TMPDIR=/tmp
python -m pytest tests -m gpu -q > gpurun_out/r3_final_gpu_tests.log 2>&1; tail -2 gpurun_out/r3_final_gpu_tests.log
python bench.py > gpurun_out/r3_final_bench.log 2> gpurun_out/r3_final_bench.err
# (1) the default timed command under --kernel-trace --stats (extras off: they are other processes / CPU work)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof -- python bench.py --no-cpu-baseline --no-other-configs --verify 0 --steps 20 --warmup 2 --batch 32 > gpurun_out/r3_bench_profiled.log 2>&1
tr=$(find gpurun_out/r3_prof -name "*kernel_trace.csv" | head -1); st=$(find gpurun_out/r3_prof -name "*kernel_stats.csv" | head -1)
python tools/profile_summary.py stats $tr $st gpurun_out/r3_bench_profiled.log --iters 80 --batch 32 > gpurun_out/r03_bench_kernel_stats.csv
rm -rf gpurun_out/r3_prof
# (2) config 5 as written (SKIP_FP16=1: the fp16 kernels did not change since the summaries under profiles/ were taken)
for B in ${SKIP_FP16:+} $( [ -z "$SKIP_FP16" ] && echo 8 32 ); do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof -- python bench.py --fp16 --depth --no-cpu-baseline --no-other-configs --verify 0 --steps 8 --warmup 2 --batch $B > gpurun_out/r3_bench_f16d_b$B.log 2>&1
  st=$(find gpurun_out/r3_prof -name "*kernel_stats.csv" | head -1); tr=$(find gpurun_out/r3_prof -name "*kernel_trace.csv" | head -1)
  head -40 $st > gpurun_out/r03_bench_fp16_rgbd_b${B}_kernel_stats.csv
  python tools/trace_iteration.py $tr 1 > gpurun_out/r03_fp16_rgbd_b${B}_iteration_trace.txt
  rm -rf gpurun_out/r3_prof
done
# (3) training iteration
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof -- python tools/bench_train.py 4 heads > gpurun_out/r3_train_prof.log 2>&1
st=$(find gpurun_out/r3_prof -name "*kernel_stats.csv" | head -1); tr=$(find gpurun_out/r3_prof -name "*kernel_trace.csv" | head -1)
head -60 $st > gpurun_out/r03_train_iteration_kernel_stats.csv
python tools/trace_train.py $tr > gpurun_out/r03_train_iteration_trace.txt
rm -rf gpurun_out/r3_prof
python tools/bench_train.py 4 heads | tail -1; python tools/bench_train.py 4 | tail -1
python tools/bench_wgrad.py 4 > gpurun_out/r03_bench_wgrad_b4.txt 2>&1; tail -1 gpurun_out/r03_bench_wgrad_b4.txt
bash tools/run_pmc_wgrad.sh > /dev/null 2>&1; head -30 gpurun_out/r3_pmc_wgrad.txt
tail -c 400 gpurun_out/r3_final_bench.log
