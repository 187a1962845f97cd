# PMC pass over the Winograd layer probe (direct, one-wave and two-wave kernels per layer): issue counters
cd /root/repo; export TMPDIR=/tmp
cd /tmp
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /root/repo/gpurun_out/r4n_pmc -- python /root/repo/tools/bench_wino.py 32 > /root/repo/gpurun_out/r4n_default.log 2>&1
cd /root/repo
f=$(find gpurun_out/r4n_pmc -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python tools/pmc_summary.py $f 2>&1 | grep -v "pack\|relayout\|build_conv" | tee gpurun_out/r04_pmc_winograd.txt
rm -rf gpurun_out/r4n_pmc
# (a second pass with TA_BUSY_avr / TA_*_STALLED_BY_TC / TCC_HIT / TCC_MISS aborts inside rocprofv3 on this image (signal 6): not collected)
