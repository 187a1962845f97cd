# kernel trace of one --heads (config 4) iteration at B = 4 and B = 32
cd /root/repo; export TMPDIR=/tmp
for B in 4 32; do
cd /tmp; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/r4_th$B -- python /root/repo/bench.py --heads --batch $B --no-cpu-baseline --no-other-configs --verify 0 --steps 4 --warmup 2 > /root/repo/gpurun_out/r4_th$B.log 2>&1
cd /root/repo
tr=$(find gpurun_out/r4_th$B -name "*kernel_trace.csv" | head -1)
python tools/trace_iteration.py $tr 1 > gpurun_out/r04_heads_b${B}_iteration_trace.txt
rm -rf gpurun_out/r4_th$B
tail -1 gpurun_out/r04_heads_b${B}_iteration_trace.txt
done
