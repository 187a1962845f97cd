# round-4 evidence run on the MI355X box (every step bounded): GPU suite, default bench line, rocprofv3 summaries for profiles/
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r4_final_gpu_tests.log 2>&1; tail -2 gpurun_out/r4_final_gpu_tests.log; fi
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r4_final_bench.log 2> gpurun_out/r4_final_bench.err; tail -c 600 gpurun_out/r4_final_bench.log; echo
# (1) the default timed command under --kernel-trace --stats
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r4_prof -- python /root/repo/bench.py --no-cpu-baseline --no-other-configs --verify 0 --steps 20 --warmup 2 --batch 32 > /root/repo/gpurun_out/r4_bench_profiled.log 2>&1; cd /root/repo
tr=$(find gpurun_out/r4_prof -name "*kernel_trace.csv" | head -1); st=$(find gpurun_out/r4_prof -name "*kernel_stats.csv" | head -1)
python tools/profile_summary.py stats $tr $st gpurun_out/r4_bench_profiled.log --iters 80 --batch 32 > gpurun_out/r04_bench_kernel_stats.csv
python tools/trace_iteration.py $tr 1 > gpurun_out/r04_b32_iteration_trace.txt
rm -rf gpurun_out/r4_prof
# (2) HBM traffic, separate PMC passes
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp; timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /root/repo/gpurun_out/r4_pmc_$c -- python /root/repo/bench.py --no-cpu-baseline --no-other-configs --verify 0 --steps 1 --warmup 1 --batch 32 > /root/repo/gpurun_out/r4_pmc_$c.log 2>&1; cd /root/repo
done
f=$(find gpurun_out/r4_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find gpurun_out/r4_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic.json
python tools/profile_summary.py traffic $f $w --iters 8 --batch 32 --key wino_B32 --md gpurun_out/r04_hbm_traffic_b32.md --json gpurun_out/hbm_traffic.json | tail -12
rm -rf gpurun_out/r4_pmc_FETCH_SIZE gpurun_out/r4_pmc_WRITE_SIZE
# (3) B = 4 iteration trace
cd /tmp; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/r4_t4 -- python /root/repo/bench.py --batch 4 --no-cpu-baseline --no-other-configs --verify 0 --steps 4 --warmup 2 > /root/repo/gpurun_out/r4_t4.log 2>&1; cd /root/repo
tr=$(find gpurun_out/r4_t4 -name "*kernel_trace.csv" | head -1)
python tools/trace_iteration.py $tr 1 > gpurun_out/r04_b4_iteration_trace.txt; tail -1 gpurun_out/r04_b4_iteration_trace.txt
rm -rf gpurun_out/r4_t4
