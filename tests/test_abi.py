"""CPU-side checks of the C-ABI boundary: the library loads and exports every symbol
include/deepim_hip.h declares (no compute calls — there is no GPU here)."""
import ctypes
import os

import pytest

from mx_deepim_amd import runtime


def test_header_parses_and_is_nonempty():
    protos = runtime.parse_header()
    assert len(protos) >= 50
    for must in ("_flow", "deepim_create", "deepim_conv2d_forward", "deepim_zoom_mask_forward",
                 "deepim_rt_transform", "deepim_transform3d_backward"):
        assert must in protos


def test_library_exports_every_declared_symbol():
    assert os.path.exists(runtime.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    dll = ctypes.CDLL(runtime.LIB_PATH)
    missing = [n for n in runtime.parse_header() if not hasattr(dll, n)]
    assert not missing, missing


def test_flow_signature_matches_reference_entry():
    # lib/flow_c/gpu_flow.hpp:1-3: (float* x6, int x4) -> void
    ret, argtypes, names = runtime.parse_header()["_flow"]
    assert ret is None
    assert names == ["flow", "valid", "depth_src", "depth_tgt", "KT", "Kinv", "batch_size", "height", "width",
                     "device_id"]
    assert argtypes[:6] == [ctypes.c_void_p] * 6 and argtypes[6:] == [ctypes.c_int] * 4


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a GPU the product path must raise, never compute on the CPU."""
    n = ctypes.c_int(-1)
    dll = runtime.lib.load()
    rc = dll.deepim_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(RuntimeError):
        runtime.Context(0)


def test_product_does_not_import_oracle():
    root = os.path.dirname(runtime._HERE)
    bad = []
    for dp, _, files in os.walk(os.path.join(root, "mx_deepim_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                if "import oracle" in src or "from oracle" in src or "oracle/" in src.replace("oracle/__init__", ""):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


CXX_FLOW = "_Z5_flowPfS_S_S_S_S_iiii"   # void _flow(float*,float*,float*,float*,float*,float*,int,int,int,int), C++ linkage


def _dynsyms(path):
    import subprocess
    out = subprocess.run(["nm", "-D", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1]: ln.split()[-2] for ln in out.splitlines() if ln.strip()}


def test_library_exports_flow_with_cxx_linkage_too():
    """lib/flow_c/gpu_flow.hpp:1-3 has no extern "C" and gpu_flow.pyx is built language="c++"
    (setup_linux.py:116-125): the reference's own binding links the mangled name."""
    syms = _dynsyms(runtime.LIB_PATH)
    assert syms.get(CXX_FLOW) == "T" and syms.get("_flow") == "T"


def test_reference_header_client_links_against_the_library(tmp_path):
    """Compile a C++ TU that includes the reference's gpu_flow.hpp unmodified and link it against libdeepim_hip.so
    with --no-undefined (build container only: /root/reference is absent on the GPU box, where the prebuilt
    oracle/_ref/libflow_hpp_client.so is called instead — tests/test_gpu_flow_reference.py)."""
    import subprocess
    hdr = "/root/reference/lib/flow_c/gpu_flow.hpp"
    root = os.path.dirname(runtime._HERE)
    if not os.path.exists(hdr):
        pytest.skip("reference checkout not present")
    out = str(tmp_path / "client.so")
    subprocess.run(["g++", "-O1", "-fPIC", "-shared", "-I" + os.path.dirname(hdr), "-o", out,
                    os.path.join(root, "oracle", "flow_hpp_client.cpp"), "-L" + os.path.dirname(runtime.LIB_PATH),
                    "-ldeepim_hip", "-Wl,--no-undefined"], check=True)
    assert _dynsyms(out).get(CXX_FLOW) == "U"
    assert "extern" not in open(hdr).read()          # the header really is C++-linkage; if that changes, so does B2
    prebuilt = os.path.join(root, "oracle", "_ref", "libflow_hpp_client.so")
    if os.path.exists(prebuilt):
        assert _dynsyms(prebuilt).get(CXX_FLOW) == "U"
