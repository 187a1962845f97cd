"""bench.py contract checks on the GPU box: the one JSON line the driver parses, at N = 1 and (rendezvous over gloo,
both ranks on the one visible GPU) the N = 2 code path incl. the per-iteration all-gather validation."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_bench_line_single_gpu():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--batch", "2", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "pose-refinement iters/sec" and d["dtype"] == "f32" and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 4 * 1 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert 0 < rf["frac"] < 1 and d["config"]["workload"]


def test_bench_two_ranks_gloo_dry_run():
    env = dict(os.environ, DEEPIM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29671", "bench.py", "--gpus", "2", "--steps", "1",
                        "--warmup", "1", "--batch", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 2 * 4 * 1 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
