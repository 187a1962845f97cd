#!/bin/bash
# evidence run of a round on the MI355X box (every step bounded): rocprofv3 summaries for profiles/ — R=06 bash tools/run_profiles.sh
# usage: R=06 bash tools/run_profiles.sh   (SKIP_TRAFFIC=1 skips the two HBM counter passes)
R=${R:-06}
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
BENCH="python /root/repo/bench.py --no-cpu-baseline --no-other-configs --no-layer-timings --verify 0 --batch 32"
prof() { d=/root/repo/gpurun_out/$1; shift; rm -rf $d; ( cd /tmp; timeout 400 rocprofv3 "$@" ) ; }
# (1) the default timed command under --kernel-trace --stats
prof tmp_prof --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/tmp_prof -- $BENCH --steps 20 --warmup 2 > gpurun_out/tmp_bench_profiled.log 2>&1
tr=$(find gpurun_out/tmp_prof -name "*kernel_trace.csv" | head -1); st=$(find gpurun_out/tmp_prof -name "*kernel_stats.csv" | head -1)
python tools/profile_summary.py stats $tr $st gpurun_out/tmp_bench_profiled.log --iters 80 --batch 32 > gpurun_out/r${R}_bench_kernel_stats.csv
python tools/trace_iteration.py $tr 1 > gpurun_out/r${R}_b32_iteration_trace.txt
grep "^{" gpurun_out/tmp_bench_profiled.log | tail -1 > gpurun_out/r${R}_bench_profiled_n1.json.log
# (2) heads graph and the training step: the H / F kernels
prof tmp_heads --kernel-trace --output-format csv -d /root/repo/gpurun_out/tmp_heads -- $BENCH --heads --steps 4 --warmup 2 > gpurun_out/tmp_heads.log 2>&1
hr=$(find gpurun_out/tmp_heads -name "*kernel_trace.csv" | head -1)
python tools/trace_iteration.py $hr 1 > gpurun_out/r${R}_heads_b32_iteration_trace.txt
prof tmp_train --kernel-trace --output-format csv -d /root/repo/gpurun_out/tmp_train -- python /root/repo/tools/bench_train.py 4 heads step4 json > gpurun_out/tmp_train.log 2>&1
tt=$(find gpurun_out/tmp_train -name "*kernel_trace.csv" | head -1)
# (2b) the 8-GPU share: one iteration at B = 4
prof tmp_b4 --kernel-trace --output-format csv -d /root/repo/gpurun_out/tmp_b4 -- python /root/repo/bench.py --no-cpu-baseline --no-other-configs --no-layer-timings --verify 0 --batch 4 --steps 4 --warmup 2 > gpurun_out/tmp_b4.log 2>&1
b4=$(find gpurun_out/tmp_b4 -name "*kernel_trace.csv" | head -1)
python tools/trace_iteration.py $b4 1 > gpurun_out/r${R}_b4_iteration_trace.txt
rm -rf gpurun_out/tmp_b4
# (3) matrix-pipe busy per dispatch of the same command (counters in their own pass)
prof tmp_pmc --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /root/repo/gpurun_out/tmp_pmc -- $BENCH --steps 1 --warmup 1 > gpurun_out/tmp_pmc.log 2>&1
pm=$(find gpurun_out/tmp_pmc -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $pm 2>&1 | grep -v "pack\|relayout\|build_conv" > gpurun_out/r${R}_pmc_bench_b32.txt
python tools/profile_summary.py perkernel $tr gpurun_out/tmp_bench_profiled.log --pmc "$pm" --heads-trace "$hr" --train-trace "$tt" --iters 80 --batch 32 --json gpurun_out/per_kernel.json > gpurun_out/r${R}_per_kernel.log 2>&1
tail -5 gpurun_out/r${R}_per_kernel.log
rm -rf gpurun_out/tmp_prof gpurun_out/tmp_heads gpurun_out/tmp_train gpurun_out/tmp_pmc
# (4) Winograd layer probe: issue counters per layer, both block shapes
prof tmp_pmcw --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /root/repo/gpurun_out/tmp_pmcw -- env WINO_PROBE=1 python /root/repo/tools/bench_wino.py 32 > gpurun_out/tmp_pmcw.log 2>&1
f=$(find gpurun_out/tmp_pmcw -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && PMC_TOL=0.25 PMC_FOLD=0 python tools/pmc_summary.py $f 2>&1 | grep -v "pack\|relayout\|build_conv" | tee gpurun_out/r${R}_pmc_winograd.txt
rm -rf gpurun_out/tmp_pmcw
# (5) HBM traffic, separate PMC passes
if [ -z "$SKIP_TRAFFIC" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  prof tmp_pmc_$c --pmc $c --kernel-trace --output-format csv -d /root/repo/gpurun_out/tmp_pmc_$c -- $BENCH --steps 1 --warmup 1 > gpurun_out/tmp_pmc_$c.log 2>&1
done
f=$(find gpurun_out/tmp_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find gpurun_out/tmp_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic.json
python tools/profile_summary.py traffic $f $w --iters 8 --batch 32 --key wino_B32 --md gpurun_out/r${R}_hbm_traffic_b32.md --json gpurun_out/hbm_traffic.json | tail -6
rm -rf gpurun_out/tmp_pmc_FETCH_SIZE gpurun_out/tmp_pmc_WRITE_SIZE
fi
