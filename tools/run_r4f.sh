# one refinement iteration of config 3's per-GPU share (B = 4, FAST_TEST graph) in launch order
cd /root/repo; export TMPDIR=/tmp
cd /tmp; rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/r4_tb4 -- python /root/repo/bench.py --batch 4 --no-cpu-baseline --no-other-configs --verify 0 --steps 6 --warmup 2 > /root/repo/gpurun_out/r4_tb4.log 2>&1
cd /root/repo
tr=$(find gpurun_out/r4_tb4 -name "*kernel_trace.csv" | head -1)
python tools/trace_iteration.py $tr 1 > gpurun_out/r04_b4_iteration_trace.txt
python tools/trace_iteration.py $tr 4 | tail -1 >> gpurun_out/r04_b4_iteration_trace.txt
rm -rf gpurun_out/r4_tb4
cat gpurun_out/r04_b4_iteration_trace.txt
