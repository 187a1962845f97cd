"""ctypes binding of libdeepim_hip.so + the device array type the operator shim works on.

The binding is generated from ``include/deepim_hip.h`` so the header stays the single
source of truth for the C ABI.  There is NO CPU fallback: if the shared library or a GPU
is missing, importing is fine but creating a ``Context`` raises.
"""
from __future__ import annotations

import ctypes
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("DEEPIM_LIB", os.path.join(_HERE, "libdeepim_hip.so"))  # override: dev ablation builds
HEADER_PATH = os.path.join(_ROOT, "include", "deepim_hip.h")


class LibraryMissing(RuntimeError):
    pass


_CTYPE = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "size_t": ctypes.c_size_t,
    "long": ctypes.c_long,
    "unsigned long long": ctypes.c_ulonglong,
    "void": None,
}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes], [argnames])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"#[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(\w+)\s*\(([^()]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef") or name in ("defined",):
            continue

        def conv(t):
            t = t.replace("const", "").strip()
            if t.endswith("**"):
                return ctypes.POINTER(ctypes.c_void_p)
            if t.endswith("*"):
                base = t[:-1].strip()
                if base == "char":
                    return ctypes.c_char_p
                if base == "int":
                    return ctypes.POINTER(ctypes.c_int)
                return ctypes.c_void_p
            return _CTYPE[t]

        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a)
                argtypes.append(conv(mm.group(1)))
                argnames.append(mm.group(2))
        protos[name] = (conv(ret), argtypes, argnames)
    return protos


class _Lib:
    """Lazy loader; attribute access returns the checked C function."""

    def __init__(self):
        self._dll = None
        self._protos = None

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise LibraryMissing(
                    "%s not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(make -C mx_deepim_amd/csrc). There is no CPU fallback." % LIB_PATH
                )
            # multi-process GPU work (RCCL between ranks) needs dmabuf IPC on this driver stack: without it hipIpcGetMemHandle fails
            # with "invalid argument". Must be in the environment before the HIP runtime starts; an explicit setting wins.
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            self._dll = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
            self._protos = parse_header()
            for name, (res, argtypes, _) in self._protos.items():
                fn = getattr(self._dll, name)  # AttributeError if the header declares a missing symbol
                fn.restype = res
                fn.argtypes = argtypes
        return self._dll

    @property
    def prototypes(self):
        self.load()
        return self._protos

    def last_error(self):
        return self.load().deepim_last_error().decode()

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        dll = self.load()
        fn = getattr(dll, name)
        if name not in self._protos or self._protos[name][0] is not ctypes.c_int or name in (
            "deepim_flow_status",
        ):
            return fn

        def checked(*args):
            rc = fn(*[_as_arg(a) for a in args])
            if rc != 0:
                raise RuntimeError("%s failed (%d): %s" % (name, rc, self.last_error()))
            return rc

        checked.__name__ = name
        return checked


def _as_arg(a):
    if isinstance(a, DeviceArray):
        return ctypes.c_void_p(a.ptr)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return ctypes.c_void_p(a.ctypes.data)
    return a


lib = _Lib()


class Context:
    """One GPU + one HIP stream (the stand-in for ``mx.gpu(i)``)."""

    _cache = {}

    def __init__(self, device_id=0):
        self.device_id = int(device_id)
        h = ctypes.c_void_p()
        lib.deepim_create(self.device_id, ctypes.byref(h))
        Context.opened.append(self.device_id)
        self.handle = h
        self.device_type = "gpu"

    @classmethod
    def get(cls, device_id=0):
        if device_id not in cls._cache:
            cls._cache[device_id] = cls(device_id)
        return cls._cache[device_id]

    # -- the process's default device: the reference's convenience calls run on "the current context"; with one process per GPU
    # (bench.py --gpus N, torch.distributed.run / torchrun launchers) that is the rank's own GPU, not GPU 0
    _default_id = None
    opened = []            # device ids this process has created contexts on, in order (the 8-rank rehearsal asserts on it)

    @classmethod
    def default_device_id(cls):
        if cls._default_id is None:
            v = os.environ.get("DEEPIM_DEVICE", os.environ.get("LOCAL_RANK", "0"))
            try:
                cls._default_id = max(0, int(v))
            except ValueError:
                cls._default_id = 0
            # launchers that mask the GPUs per rank (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES, SLURM --gpus-per-task) leave every
            # rank ONE visible device and LOCAL_RANK > 0: fold the rank onto the visible devices, as bench.py does
            n = cls.visible_devices()
            if n > 0 and cls._default_id >= n and "DEEPIM_DEVICE" not in os.environ:
                cls._default_id %= n
        return cls._default_id

    @staticmethod
    def visible_devices():
        """hipGetDeviceCount through the library; 0 without a GPU or without the library (nothing is created)."""
        try:
            n = ctypes.c_int(0)
            lib.load().deepim_device_count(ctypes.byref(n))
            return int(n.value)
        except Exception:       # noqa: BLE001
            return 0

    @classmethod
    def set_default(cls, device_id):
        cls._default_id = int(device_id)

    @classmethod
    def default(cls):
        """Context of the process's default device: DEEPIM_DEVICE, else LOCAL_RANK (one process per GPU), else 0."""
        return cls.get(cls.default_device_id())

    def __repr__(self):
        return "gpu(%d)" % self.device_id

    # -- memory
    def empty(self, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype=np.float32):
        a = DeviceArray(self, shape, dtype)
        lib.deepim_memset(self.handle, a, 0, a.nbytes)
        return a

    def array(self, host, dtype=np.float32):
        host = np.ascontiguousarray(host, dtype=dtype)
        a = DeviceArray(self, host.shape, dtype)
        if a.nbytes:
            lib.deepim_h2d(self.handle, a, host, a.nbytes)
        return a

    def sync(self):
        lib.deepim_sync(self.handle)

    # -- HIP-event stopwatch on the context stream
    def timer(self):
        t = ctypes.c_int()
        lib.deepim_timer_create(self.handle, ctypes.byref(t))
        return _Timer(self, t.value)


class _Timer:
    def __init__(self, ctx, tid):
        self.ctx, self.tid = ctx, tid

    def start(self):
        lib.deepim_timer_start(self.ctx.handle, self.tid)

    def stop(self):
        lib.deepim_timer_stop(self.ctx.handle, self.tid)

    def elapsed_ms(self):
        ms = ctypes.c_float()
        lib.load().deepim_timer_elapsed_ms(self.ctx.handle, self.tid, ctypes.byref(ms))
        return ms.value


class DeviceArray:
    """Dense C-contiguous device tensor (the NDArray stand-in). Owns its allocation unless a view."""

    def __init__(self, ctx, shape, dtype=np.float32, ptr=None, base=None):
        self.context = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape)) if self.shape else 1
        self.nbytes = self.size * self.dtype.itemsize
        self._base = base
        if ptr is None:
            p = ctypes.c_void_p()
            lib.deepim_malloc(ctx.handle, max(self.nbytes, 4), ctypes.byref(p))
            self.ptr = p.value
            self._owner = True
        else:
            self.ptr = int(ptr)
            self._owner = False

    def __del__(self):
        try:
            if getattr(self, "_owner", False) and self.ptr:
                lib.load().deepim_free(self.context.handle, ctypes.c_void_p(self.ptr))
                self.ptr = 0
        except Exception:
            pass

    @property
    def ndim(self):
        return len(self.shape)

    def asnumpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            lib.deepim_d2h(self.context.handle, out, self, self.nbytes)
        return out

    def reshape(self, shape):
        shape = tuple(shape)
        if -1 in shape:
            known = int(np.prod([s for s in shape if s != -1]))
            shape = tuple(self.size // known if s == -1 else s for s in shape)
        assert int(np.prod(shape)) == self.size
        return DeviceArray(self.context, shape, self.dtype, ptr=self.ptr, base=self)

    def __getitem__(self, idx):
        """Leading-axis view (``a[i]`` / ``a[i:j]``) — enough for the operator shim."""
        inner = int(np.prod(self.shape[1:])) if len(self.shape) > 1 else 1
        if isinstance(idx, slice):
            start, stop, step = idx.indices(self.shape[0])
            assert step == 1
            return DeviceArray(self.context, (stop - start,) + self.shape[1:], self.dtype,
                               ptr=self.ptr + start * inner * self.dtype.itemsize, base=self)
        i = int(idx)
        if i < 0:
            i += self.shape[0]
        return DeviceArray(self.context, self.shape[1:], self.dtype,
                           ptr=self.ptr + i * inner * self.dtype.itemsize, base=self)

    def copyfrom(self, src):
        if isinstance(src, DeviceArray):
            assert src.nbytes == self.nbytes
            lib.deepim_d2d(self.context.handle, self, src, self.nbytes)
        else:
            host = np.ascontiguousarray(np.broadcast_to(np.asarray(src, dtype=self.dtype), self.shape))
            lib.deepim_h2d(self.context.handle, self, host, self.nbytes)
        return self

    def copy(self):
        out = DeviceArray(self.context, self.shape, self.dtype)
        lib.deepim_d2d(self.context.handle, out, self, self.nbytes)
        return out

    def __repr__(self):
        return "<DeviceArray %s %s @%s>" % (self.shape, self.dtype, self.context)
