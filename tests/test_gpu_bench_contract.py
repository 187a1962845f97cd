"""bench.py contract checks on the GPU box: the one JSON line the driver parses, at N = 1 and — launched exactly as the
driver launches it (torch.distributed.run) with both ranks on the one visible GPU — the N = 2 code path incl. the
per-iteration all-gather validation, weak and strong (--global-batch) scaling.  RCCL refuses two ranks on one device, so
the two-rank runs route the same exchange through the host rendezvous (DEEPIM_BENCH_BACKEND=host); the RCCL entry points
themselves are exercised with a world-size-1 communicator."""
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


def _run_two_ranks(port, extra):
    env = dict(os.environ, DEEPIM_BENCH_BACKEND="host", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "1",
                        "--warmup", "1"] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return _last_json(r.stdout)


def test_bench_two_ranks_weak_scaling_dry_run():
    d = _run_two_ranks(29671, ["--batch", "2"])
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 2 * 4 * 1 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_bench_two_ranks_strong_scaling_ragged_dry_run():
    """--global-batch 3 over 2 ranks: shards of 2 and 1 pairs (padded all-gather), value counts the 3 global pairs."""
    d = _run_two_ranks(29675, ["--global-batch", "3"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 3
    assert abs(d["value"] - 3 * 4 * 1 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_rccl_entry_points_world_size_one(ctx):
    """deepim_comm_unique_id / comm_init / allgather_poses / comm_allreduce_f64 / comm_destroy through librccl.so itself
    (dlopen'ed by the library) with a one-rank communicator, plus the communicator-less copy path."""
    import ctypes
    import numpy as np
    from mx_deepim_amd.runtime import lib
    rng = np.random.default_rng(3)
    poses = rng.standard_normal((5, 3, 4)).astype(np.float32)
    src, dst = ctx.array(poses), ctx.zeros((5, 3, 4))
    lib.deepim_allgather_poses(ctx.handle, dst, src, 5)            # no communicator: plain copy
    np.testing.assert_array_equal(dst.asnumpy(), poses)
    uid = ctypes.create_string_buffer(128)
    lib.deepim_comm_unique_id(uid)
    assert any(uid.raw)
    lib.deepim_comm_init(ctx.handle, 0, 1, uid)
    try:
        dst2 = ctx.zeros((5, 3, 4))
        lib.deepim_allgather_poses(ctx.handle, dst2, src, 5)       # ncclAllGather on the library's stream
        t = ctx.array(np.array([1.25, -3.0]), dtype=np.float64)
        lib.deepim_comm_allreduce_f64(ctx.handle, t, 2, 0)
        np.testing.assert_array_equal(dst2.asnumpy(), poses)
        np.testing.assert_array_equal(t.asnumpy(), [1.25, -3.0])
        with __import__("pytest").raises(RuntimeError):
            lib.deepim_comm_init(ctx.handle, 0, 1, uid)             # already initialised
    finally:
        lib.deepim_comm_destroy(ctx.handle)
