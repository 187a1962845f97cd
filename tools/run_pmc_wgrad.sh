cd /root/repo; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r3_pmc_wg -- python tools/bench_wgrad.py 4 > gpurun_out/r3_pmc_wg.log 2>&1
f=$(find gpurun_out/r3_pmc_wg -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $f > gpurun_out/r3_pmc_wgrad.txt 2>&1
rm -rf gpurun_out/r3_pmc_wg
cat gpurun_out/r3_pmc_wgrad.txt
