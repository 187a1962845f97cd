"""Reader / writer for MXNet's NDArray-list container (the `.params` checkpoints `mx.nd.load` / `mx.nd.save`
exchange; the reference loads them at lib/utils/load_model.py:21).

MXNet is not installed and no checkpoint exists offline, so the layout below is RESTATED from MXNet 1.2's
`src/ndarray/ndarray.cc` (`NDArray::Save/Load`, list magic 0x112) and exercised by round trips only — it is
not pinned against a file written by MXNet itself. Little-endian throughout:

    uint64 0x112 | uint64 reserved(0)
    uint64 n_arrays, then per array
        V2: uint32 0xF993FAC9 | int32 storage_type(0 = dense) | shape | ctx | int32 type_flag | raw data
        V1: uint32 0xF993FAC8 |                                 shape | ctx | int32 type_flag | raw data
        legacy (no magic): uint32 ndim | uint32 dims[ndim]    | ctx | int32 type_flag | raw data
      shape = uint32 ndim | int64 dims[ndim];   ctx = int32 dev_type | int32 dev_id;   ndim 0 = empty array, no payload
    uint64 n_names, then per name: uint64 length | bytes
"""
import struct

import numpy as np

LIST_MAGIC = 0x112
V1_MAGIC, V2_MAGIC = 0xF993FAC8, 0xF993FAC9
# mshadow type flags
TYPE_FLAGS = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
FLAG_OF = {np.dtype(v): k for k, v in TYPE_FLAGS.items()}


class _Reader(object):
    def __init__(self, buf):
        self.buf, self.pos = memoryview(buf), 0

    def take(self, fmt):
        size = struct.calcsize(fmt)
        if self.pos + size > len(self.buf):
            raise ValueError("truncated NDArray file")
        out = struct.unpack_from(fmt, self.buf, self.pos)
        self.pos += size
        return out if len(out) > 1 else out[0]

    def raw(self, n):
        if self.pos + n > len(self.buf):
            raise ValueError("truncated NDArray file")
        out = self.buf[self.pos:self.pos + n]
        self.pos += n
        return out


def _read_array(r):
    magic = r.take("<I")
    if magic == V2_MAGIC:
        stype = r.take("<i")
        if stype != 0:
            raise NotImplementedError("sparse NDArray (storage type %d) in checkpoint" % stype)
    if magic in (V1_MAGIC, V2_MAGIC):
        ndim = r.take("<I")
        shape = tuple(np.frombuffer(r.raw(8 * ndim), dtype="<i8").tolist())
    else:                      # legacy: the word just read is ndim, dims are uint32
        ndim = magic
        if ndim > 32:
            raise ValueError("not an NDArray record (bad magic 0x%08x)" % magic)
        shape = tuple(np.frombuffer(r.raw(4 * ndim), dtype="<u4").tolist())
    if ndim == 0:
        return None
    r.take("<ii")              # saved context: ignored, arrays land on the host
    flag = r.take("<i")
    if flag not in TYPE_FLAGS:
        raise ValueError("unknown type flag %d" % flag)
    dtype = np.dtype(TYPE_FLAGS[flag]).newbyteorder("<")
    count = int(np.prod(shape, dtype=np.int64))
    return np.frombuffer(r.raw(count * dtype.itemsize), dtype=dtype).reshape(shape).astype(dtype.newbyteorder("="))


def load(fname):
    """-> dict name -> numpy array (or a list when the file carries no names), like `mx.nd.load`."""
    with open(fname, "rb") as fh:
        r = _Reader(fh.read())
    magic, _ = r.take("<QQ")
    if magic != LIST_MAGIC:
        raise ValueError("%s: not an NDArray list file (magic 0x%x)" % (fname, magic))
    arrays = [_read_array(r) for _ in range(r.take("<Q"))]
    names = []
    for _ in range(r.take("<Q")):
        names.append(bytes(r.raw(r.take("<Q"))).decode("utf-8"))
    if not names:
        return arrays
    if len(names) != len(arrays):
        raise ValueError("%s: %d names for %d arrays" % (fname, len(names), len(arrays)))
    return dict(zip(names, arrays))


def save(fname, data):
    """`mx.nd.save`: dict name -> array, or a list of arrays. Always writes V2 dense records."""
    if isinstance(data, dict):
        names, arrays = list(data.keys()), list(data.values())
    else:
        names, arrays = [], list(data)
    out = [struct.pack("<QQ", LIST_MAGIC, 0), struct.pack("<Q", len(arrays))]
    for a in arrays:
        a = np.ascontiguousarray(a.asnumpy() if hasattr(a, "asnumpy") else a)
        if a.dtype not in FLAG_OF:
            raise TypeError("dtype %s has no NDArray type flag" % a.dtype)
        out.append(struct.pack("<Ii", V2_MAGIC, 0))
        out.append(struct.pack("<I", a.ndim) + np.asarray(a.shape, dtype="<i8").tobytes())
        if a.ndim == 0:
            continue
        out.append(struct.pack("<iii", 1, 0, FLAG_OF[a.dtype]))      # cpu(0)
        out.append(a.astype(a.dtype.newbyteorder("<")).tobytes())
    out.append(struct.pack("<Q", len(names)))
    for n in names:
        b = n.encode("utf-8")
        out.append(struct.pack("<Q", len(b)) + b)
    with open(fname, "wb") as fh:
        fh.write(b"".join(out))
