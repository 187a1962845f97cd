"""`.params` import: NDArray-list container round trips and the load_model.py mirror.
No MXNet-written file exists offline, so the byte layout is additionally checked against a record assembled by hand
from the documented field order."""
import struct

import numpy as np
import pytest

from mx_deepim_amd.lib.utils import ndarray_file
from mx_deepim_amd.lib.utils.load_model import load_checkpoint, load_param, save_checkpoint


def test_round_trip_dict_and_list(tmp_path):
    rng = np.random.default_rng(0)
    blob = {"arg:flow_conv1_weight": rng.standard_normal((64, 6, 7, 7)).astype(np.float32),
            "arg:fc6_bias": rng.standard_normal(256).astype(np.float32),
            "aux:bn_moving_var": rng.random(3).astype(np.float64),
            "arg:half": rng.standard_normal((2, 3)).astype(np.float16),
            "arg:ids": np.arange(5, dtype=np.int32)}
    f = str(tmp_path / "x.params")
    ndarray_file.save(f, blob)
    back = ndarray_file.load(f)
    assert list(back) == list(blob)
    for k in blob:
        assert back[k].dtype == blob[k].dtype
        np.testing.assert_array_equal(back[k], blob[k])
    ndarray_file.save(f, [blob["arg:fc6_bias"], blob["arg:ids"]])
    lst = ndarray_file.load(f)
    assert isinstance(lst, list) and len(lst) == 2
    np.testing.assert_array_equal(lst[1], blob["arg:ids"])


def _hand_record(version, a):
    if version == 2:
        head = struct.pack("<Ii", 0xF993FAC9, 0) + struct.pack("<I", a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape)
    elif version == 1:
        head = struct.pack("<I", 0xF993FAC8) + struct.pack("<I", a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape)
    else:
        head = struct.pack("<I", a.ndim) + struct.pack("<%dI" % a.ndim, *a.shape)
    return head + struct.pack("<ii", 2, 3) + struct.pack("<i", 0) + a.tobytes()     # saved on gpu(3), float32


@pytest.mark.parametrize("version", [2, 1, 0])
def test_reads_all_record_versions(tmp_path, version):
    a = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    name = b"arg:rot_weight"
    raw = struct.pack("<QQ", 0x112, 0) + struct.pack("<Q", 1) + _hand_record(version, a) + \
        struct.pack("<Q", 1) + struct.pack("<Q", len(name)) + name
    f = tmp_path / "h.params"
    f.write_bytes(raw)
    back = ndarray_file.load(str(f))
    np.testing.assert_array_equal(back["arg:rot_weight"], a)
    if version == 2:                                  # the writer emits exactly this layout (cpu(0) instead of gpu(3))
        g = str(tmp_path / "w.params")
        ndarray_file.save(g, {"arg:rot_weight": a})
        mine = open(g, "rb").read()
        assert mine[:16 + 8 + 8 + 4 + 24] == raw[:16 + 8 + 8 + 4 + 24]
        assert mine[-(8 + 8 + len(name)):] == raw[-(8 + 8 + len(name)):]
        assert len(mine) == len(raw)


def test_rejects_garbage(tmp_path):
    f = tmp_path / "bad.params"
    f.write_bytes(b"\x00" * 40)
    with pytest.raises(ValueError):
        ndarray_file.load(str(f))
    ok = tmp_path / "ok.params"
    ndarray_file.save(str(ok), {"arg:w": np.ones((4, 4), np.float32)})
    data = ok.read_bytes()
    f.write_bytes(data[:-20])
    with pytest.raises(ValueError):
        ndarray_file.load(str(f))


def test_load_param_mirror_and_flownet_import(tmp_path):
    from mx_deepim_amd.config import default_config
    from mx_deepim_amd.symbols import deepIM_flownet
    cfg = default_config()
    net = deepIM_flownet().get_symbol(cfg)
    shapes = net.arg_shape_dict()
    rng = np.random.default_rng(1)
    # a 6-channel FlowNet-style checkpoint with the suffixes older files carry
    ckpt = {k: rng.standard_normal(shapes[k]).astype(np.float32)
            for k in ("flow_conv1_bias", "conv2_weight", "conv2_bias", "conv3_bias")}
    c1 = shapes["flow_conv1_weight"]
    ckpt["flow_conv1_weight"] = rng.standard_normal((c1[0], 6) + tuple(c1[2:])).astype(np.float32)
    ckpt["fc7_weight_test"] = rng.standard_normal(shapes["fc7_weight"]).astype(np.float32)
    prefix = str(tmp_path / "flownet")
    save_checkpoint(prefix, 7, ckpt, {"dummy_moving_mean": np.zeros(2, np.float32)})
    arg, aux = load_checkpoint(prefix, 7)
    assert set(arg) == set(ckpt) and list(aux) == ["dummy_moving_mean"]
    arg, aux = load_param(prefix, 7, process=True)
    assert "fc7_weight" in arg and "fc7_weight_test" not in arg
    merged = deepIM_flownet.adapt_checkpoint(dict(arg), shapes)     # what init_weights applies before filling the rest
    assert merged["flow_conv1_weight"].shape == tuple(c1)
    np.testing.assert_array_equal(merged["flow_conv1_weight"][:, :6], ckpt["flow_conv1_weight"])
    assert not merged["flow_conv1_weight"][:, 6:].any()
    np.testing.assert_array_equal(merged["conv2_weight"], ckpt["conv2_weight"])
    np.testing.assert_array_equal(merged["fc7_weight"], ckpt["fc7_weight_test"])
    assert set(merged) <= set(shapes) and all(tuple(merged[k].shape) == tuple(shapes[k]) for k in merged)
    with pytest.raises((IOError, OSError)):
        load_checkpoint(prefix, 8)
