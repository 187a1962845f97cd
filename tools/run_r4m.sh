cd /root/repo
python bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 > gpurun_out/r4m_b32.json 2> gpurun_out/r4m_b32.err; python - <<'PY'
import json
for f in ("r4m_b32",):
    j=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, j["value"], j["roofline"]["achieved"], j["roofline"].get("executed"), j.get("parity"))
PY
python bench.py --no-winograd --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 --verify 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-wino B32', j['value'], j['roofline']['achieved'])"
python bench.py --batch 4 --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B4', j['value'], j['roofline']['achieved'], j['roofline'].get('winograd_layers'), j.get('parity'))"
python bench.py --batch 8 --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --verify 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B8', j['value'], j['roofline']['achieved'], j['roofline'].get('winograd_layers'))"
python bench.py --batch 8 --no-winograd --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --verify 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B8 no-wino', j['value'], j['roofline']['achieved'])"
for b in 8 16; do timeout 200 python tools/bench_wino.py $b | tail -2; done
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15
