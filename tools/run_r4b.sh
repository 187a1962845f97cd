#!/bin/bash
# round 4, GPU session B: where the ping-pong kernel's time goes — ablation variants, per-phase timeline, PMC counters
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_backward.py tests/test_gpu_render.py -q -k "train_step or lit_render" 2>&1 | tail -15 ) > gpurun_out/r4b_tests.log
for v in abl1 abl2 abl4 abl8 abl16; do
  ( echo "== $v"; DEEPIM_LIB=variants/lib_$v.so timeout 120 python tools/bench_layers_f16.py 32 8 0 2>&1 | grep -E "conv2|conv3|conv4|encoder" ) >> gpurun_out/r4b_ablations.log
done
( DEEPIM_LIB=variants/lib_trace.so timeout 120 python tools/pp_trace.py 2 32 2>&1 | tail -40 ) > gpurun_out/r4b_trace_conv3.log
( DEEPIM_LIB=variants/lib_trace.so timeout 120 python tools/pp_trace.py 5 32 2>&1 | tail -40 ) > gpurun_out/r4b_trace_conv4_1.log
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /root/repo/gpurun_out/r4b_pmc1 -- python /root/repo/tools/bench_layers_f16.py 32 8 0,16 > /root/repo/gpurun_out/r4b_pmc1.log 2>&1
cd /root/repo
f=$(find gpurun_out/r4b_pmc1 -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $f > gpurun_out/r4b_pmc1_summary.txt 2>&1
rm -rf gpurun_out/r4b_pmc1
( rocprofv3 -L 2>&1 | grep -oE "\b(TCP|TCC|TA|TD|SQ|SQC)_[A-Za-z0-9_]+" | sort -u | tr '\n' ' ' ) > gpurun_out/r4b_counters.txt
cd /tmp
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /root/repo/gpurun_out/r4b_pmc2 -- python /root/repo/tools/bench_layers_f16.py 32 8 0,16 > /root/repo/gpurun_out/r4b_pmc2.log 2>&1
cd /root/repo
f=$(find gpurun_out/r4b_pmc2 -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" > gpurun_out/r4b_pmc2_summary.txt 2>&1 <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    if "conv_f16" not in r["Kernel_Name"]: continue
    k = (r["Kernel_Name"][:70], r.get("Grid_Size"))
    d.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in d.items():
    print(k, {a: "%.4g" % b for a, b in v.items()})
PY
rm -rf gpurun_out/r4b_pmc2
tail -3 gpurun_out/r4b_tests.log; cat gpurun_out/r4b_ablations.log; cat gpurun_out/r4b_trace_conv3.log | tail -12; cat gpurun_out/r4b_pmc1_summary.txt
