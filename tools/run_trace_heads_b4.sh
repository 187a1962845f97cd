# one refinement iteration of config 4's per-GPU share (B = 4, decoder + mask / flow heads) in launch order
cd /root/repo; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3_th -- python bench.py --heads --batch 4 --no-cpu-baseline --no-other-configs --verify 0 --steps 6 --warmup 2 > gpurun_out/r3_th.log 2>&1
tr=$(find gpurun_out/r3_th -name "*kernel_trace.csv" | head -1)
python tools/trace_iteration.py $tr 1 > gpurun_out/r03_heads_b4_iteration_trace.txt
rm -rf gpurun_out/r3_th
tail -3 gpurun_out/r03_heads_b4_iteration_trace.txt
