for fs in 0 32 96 128 192 256; do timeout 300 python bench.py --batch 4 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --verify 0 --opt fc_slices=$fs 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fc_slices=$fs B4', round(j['value'],1), 'it/s')"; done
python tools/bench_fc6.py 2>&1 | tail -12
