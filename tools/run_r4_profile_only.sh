cd /root/repo; export TMPDIR=/tmp
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r4_prof -- python /root/repo/bench.py --no-cpu-baseline --no-other-configs --verify 0 --steps 20 --warmup 2 --batch 32 > /root/repo/gpurun_out/r4_bench_profiled.log 2>&1; cd /root/repo
tr=$(find gpurun_out/r4_prof -name "*kernel_trace.csv" | head -1); st=$(find gpurun_out/r4_prof -name "*kernel_stats.csv" | head -1)
python tools/profile_summary.py stats $tr $st gpurun_out/r4_bench_profiled.log --iters 80 --batch 32 > gpurun_out/r04_bench_kernel_stats.csv
python tools/trace_iteration.py $tr 1 > gpurun_out/r04_b32_iteration_trace.txt
rm -rf gpurun_out/r4_prof
grep "conv launch group" gpurun_out/r04_bench_kernel_stats.csv
