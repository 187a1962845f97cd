# HBM traffic of the default configuration from PMC counters, separate passes (FETCH_SIZE, WRITE_SIZE), as the guide prescribes
cd /root/repo; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r3_pmc_$c -- python bench.py --no-cpu-baseline --no-other-configs --verify 0 --steps 1 --warmup 1 --batch 32 > gpurun_out/r3_pmc_$c.log 2>&1
done
f=$(find gpurun_out/r3_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find gpurun_out/r3_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic.json
python tools/profile_summary.py traffic $f $w --iters 8 --batch 32 --md gpurun_out/r03_hbm_traffic_b32.md --json gpurun_out/hbm_traffic.json | tail -12
rm -rf gpurun_out/r3_pmc_FETCH_SIZE gpurun_out/r3_pmc_WRITE_SIZE
