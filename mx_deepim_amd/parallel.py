"""Multi-GPU layout of the refinement loop, without PyTorch.

Pairs are independent units, sharded in contiguous blocks across ranks (one process per GPU); weights are
replicated; the only exchange on the data path is ONE all-gather of the refined (B_local,3,4) float32 poses per
refinement iteration — `deepim_allgather_poses` = `ncclAllGather` (RCCL over xGMI) enqueued on the library's own
stream, no host sync inside the loop.  The reference's counterpart is the per-device executor group merging the
outputs of all GPUs for the host (deepim/core/DataParallelExecutorGroup.py:364-388, deepim/test.py:135).

Host side:
  * `Rendezvous` — a tiny TCP star (rank 0 listens on MASTER_ADDR:MASTER_PORT+1 — 127.0.0.1 for a single-node job —
    the others connect) used to ship the 128-byte RCCL unique id at start-up and for the few host-side collectives a
    launcher needs (barrier, all-gather of small typed values, max-over-ranks of a timing).  Typed binary frames (no
    pickle), HMAC-authenticated membership.  It reads the same environment variables
    `python -m torch.distributed.run` sets (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT) but does not
    import torch.  Works without a GPU, which is how the world_size-2 CPU tests exercise it.
  * `PoseComm` — the RCCL communicator of a `Context`: `init` (bootstrap through a Rendezvous), `all_gather_poses`
    (device → device, asynchronous), `max_over_ranks` (device all-reduce of a float64).
  * `shard_bounds` / `shard_pairs` — the partition itself.
"""
import ctypes
import hashlib
import hmac
import os
import socket
import struct
import time

import numpy as np


def shard_bounds(n_pairs, world_size, rank):
    """Contiguous block partition; the first (n_pairs % world_size) ranks take one extra pair."""
    base, extra = divmod(int(n_pairs), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_counts(n_pairs, world_size):
    return [shard_bounds(n_pairs, world_size, r)[1] - shard_bounds(n_pairs, world_size, r)[0] for r in range(world_size)]


def shard_pairs(batch, world_size, rank, n_pairs=None, replicated=("K",)):
    """Slice every per-pair array of a batch dict (leading axis = pairs; frame-major arrays have the pair
    axis second) to this rank's block. Keys in `replicated` (the intrinsics) are handed to every rank whole."""
    n_pairs = n_pairs if n_pairs is not None else batch["image_observed"].shape[0]
    lo, hi = shard_bounds(n_pairs, world_size, rank)
    out = {}
    for k, v in batch.items():
        a = np.asarray(v)
        if k in replicated:
            out[k] = v
        elif a.ndim >= 1 and a.shape[0] == n_pairs:
            out[k] = a[lo:hi]
        elif a.ndim >= 2 and a.shape[1] == n_pairs:
            out[k] = a[:, lo:hi]
        else:
            out[k] = v
    return out


# ----------------------------------------------------------------------------------------------- rendezvous ----
# Wire format: fixed binary framing, no pickle — a peer can only ever hand us one of the five value kinds below, and
# nothing on the wire is executable.  frame = <u64 length><payload>;  payload = <u8 kind> + body:
#   N  none                      F  <f64>                  I  <i64>                  B  raw bytes
#   A  <u8 dtype code><u8 ndim><u32 shape[ndim]> raw C-order data           L  <u32 count> then `count` nested frames
_MAX_FRAME = 64 << 20           # control plane only: ids, timings, (B,3,4) poses
_DTYPES = {0: np.dtype("<f4"), 1: np.dtype("<f8"), 2: np.dtype("<i4"), 3: np.dtype("<i8"), 4: np.dtype("u1")}
_DTYPE_CODE = {v: k for k, v in _DTYPES.items()}


def _encode(obj):
    if obj is None:
        return b"N"
    if isinstance(obj, (bytes, bytearray)):
        return b"B" + bytes(obj)
    if isinstance(obj, (bool, int, np.integer)):
        return b"I" + struct.pack("<q", int(obj))
    if isinstance(obj, (float, np.floating)):
        return b"F" + struct.pack("<d", float(obj))
    if isinstance(obj, np.ndarray):
        a = np.ascontiguousarray(obj)
        dt = a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else a.dtype
        if dt not in _DTYPE_CODE:
            raise TypeError("rendezvous: unsupported array dtype %s" % a.dtype)
        if a.ndim > 8:
            raise TypeError("rendezvous: arrays of at most 8 dimensions")
        return (b"A" + struct.pack("<BB", _DTYPE_CODE[dt], a.ndim) + struct.pack("<%dI" % a.ndim, *a.shape) +
                a.astype(dt, copy=False).tobytes())
    if isinstance(obj, (list, tuple)):
        parts = [_encode(o) for o in obj]
        return b"L" + struct.pack("<I", len(parts)) + b"".join(struct.pack("<Q", len(p)) + p for p in parts)
    raise TypeError("rendezvous: cannot send %r (None, int, float, bytes, ndarray or a list of those)" % type(obj))


def _decode(buf):
    if not buf:
        raise ValueError("rendezvous: empty frame")
    kind, body = buf[:1], memoryview(buf)[1:]
    if kind == b"N" and len(body) == 0:
        return None
    if kind == b"B":
        return bytes(body)
    if kind == b"I" and len(body) == 8:
        return struct.unpack("<q", body)[0]
    if kind == b"F" and len(body) == 8:
        return struct.unpack("<d", body)[0]
    if kind == b"A" and len(body) >= 2:
        code, ndim = struct.unpack("<BB", body[:2])
        if code in _DTYPES and ndim <= 8 and len(body) >= 2 + 4 * ndim:
            shape = struct.unpack("<%dI" % ndim, body[2:2 + 4 * ndim])
            data = body[2 + 4 * ndim:]
            n = 1
            for d in shape:
                n *= d
            if n * _DTYPES[code].itemsize == len(data):
                return np.frombuffer(data, dtype=_DTYPES[code]).reshape(shape).copy()
    if kind == b"L" and len(body) >= 4:
        (count,) = struct.unpack("<I", body[:4])
        out, off = [], 4
        for _ in range(count):
            if off + 8 > len(body):
                break
            (ln,) = struct.unpack("<Q", body[off:off + 8])
            off += 8
            if ln > len(body) - off:
                break
            out.append(_decode(bytes(body[off:off + ln])))
            off += ln
        if len(out) == count and off == len(body):
            return out
    raise ValueError("rendezvous: malformed frame (kind %r, %d bytes)" % (kind, len(buf)))


def _send_msg(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv_msg(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > _MAX_FRAME:
        raise ValueError("rendezvous: frame of %d bytes exceeds the %d-byte limit" % (n, _MAX_FRAME))
    return _recv_exact(sock, n)


def _is_loopback(addr):
    return addr in ("localhost", "::1") or addr.startswith("127.")


class Rendezvous(object):
    """TCP star between the ranks of one job: rank 0 is the hub.  Every collective is gather-to-hub + broadcast;
    payloads are small (ids, timings), this is control plane only.

    Trust: frames are decoded by `_decode` (typed binary, nothing executable).  Membership is authenticated with an
    HMAC-SHA256 challenge in both directions over a job token (`DEEPIM_RDZV_TOKEN`, or the `token` argument).  A
    single-node job (the supported deployment: WORLD_SIZE == LOCAL_WORLD_SIZE, or a loopback MASTER_ADDR) binds and
    connects on 127.0.0.1 only and may run without an explicit token; a routable MASTER_ADDR REQUIRES the token."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0, token=None):
        env = os.environ
        self.rank = int(env.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(env.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.addr = addr or env.get("MASTER_ADDR", "127.0.0.1")
        # MASTER_PORT itself belongs to the launcher's own store; the job's rendezvous sits one above it
        self.port = int(env.get("DEEPIM_RDZV_PORT", int(env.get("MASTER_PORT", "29500")) + 1)) if port is None else int(port)
        self.port0 = self.port
        self.timeout = float(timeout)
        self.handshake_timeout = 2.0     # hub: per-connection limit for the 52-byte hello
        # spoke: how long to keep scanning after some hub REFUSED this rank (see _connect) — a share of the overall timeout, so that a
        # rank 0 that binds its port late on a busy host is still found (a fixed 5 s failed such jobs spuriously)
        self.reject_grace = max(5.0, 0.5 * self.timeout)
        single_node = int(env.get("LOCAL_WORLD_SIZE", "0")) == self.world or _is_loopback(self.addr)
        if single_node:
            self.addr = "127.0.0.1"
        tok = token if token is not None else env.get("DEEPIM_RDZV_TOKEN")
        if tok is None:
            if not single_node and self.world > 1:
                raise RuntimeError("rendezvous over a routable address (%s) needs a shared secret: set DEEPIM_RDZV_TOKEN "
                                   "to the same random string on every rank" % self.addr)
            # loopback only: the launcher's run id separates concurrent jobs of one host; it is not a secret
            tok = "deepim-local:%s:%d:%d" % (env.get("TORCHELASTIC_RUN_ID", "none"), self.port, self.world)
        self._key = hashlib.sha256(tok.encode() if isinstance(tok, str) else bytes(tok)).digest()
        self._peers = {}      # hub: rank -> socket
        self._hub = None      # spoke: socket to rank 0
        self._listener = None
        if self.world > 1:
            self._connect()

    def _candidates(self):
        """Ports the hub may listen on, in order: the configured one (MASTER_PORT + 1 / DEEPIM_RDZV_PORT) and five fallbacks."""
        return [self.port0 + 97 * k for k in range(6) if self.port0 + 97 * k < 65536]

    def _mac(self, *parts):
        return hmac.new(self._key, b"|".join(parts), hashlib.sha256).digest()

    def _connect(self):
        if self.rank == 0:
            ls, err = None, None
            for port in self._candidates():      # MASTER_PORT + 1 may be somebody else's: the spokes scan the same short list
                ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    ls.bind((self.addr, port))
                    self.port = port
                    break
                except OSError as e:
                    ls.close()
                    ls, err = None, e
            if ls is None:
                raise RuntimeError("rendezvous: cannot bind any of %s on %s (%s) — set DEEPIM_RDZV_PORT to a free port on every "
                                   "rank" % (self._candidates(), self.addr, err))
            ls.listen(self.world)
            ls.settimeout(self.timeout)
            self._listener = ls
            deadline = time.time() + self.timeout
            while len(self._peers) < self.world - 1:
                if time.time() > deadline:
                    raise RuntimeError("rendezvous: only %d of %d ranks joined" % (len(self._peers) + 1, self.world))
                try:
                    ls.settimeout(max(0.05, min(1.0, deadline - time.time())))
                    conn, _ = ls.accept()
                except socket.timeout:
                    continue                   # the deadline check above words the failure
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(min(self.timeout, self.handshake_timeout))   # a silent stranger stalls the joining ranks this long at most
                try:            # a stranger (wrong token, bad rank, garbage) is dropped; it cannot take a rank's slot
                    nonce = os.urandom(16)
                    conn.sendall(nonce)
                    hello = _recv_exact(conn, 4 + 16 + 32)      # rank, the spoke's own nonce, HMAC(nonce | rank)
                    (r,) = struct.unpack("<I", hello[:4])
                    ok = hmac.compare_digest(hello[20:], self._mac(b"spoke", nonce, hello[:4]))
                    if not ok or r <= 0 or r >= self.world or r in self._peers:
                        raise ValueError("rejected")
                    conn.sendall(self._mac(b"hub", hello[4:20]))   # prove the hub knows the token too
                except (ValueError, OSError, ConnectionError):
                    conn.close()
                    continue
                conn.settimeout(self.timeout)
                self._peers[r] = conn
        else:
            # the hub listens on the first port of the candidate list it could bind: try them in turn until one answers the
            # challenge with this job's token (a foreign service on a candidate port fails the handshake and is skipped)
            # A refusal (wrong token / duplicate rank / failed proof) does not end the scan: when another deepim job of this host
            # owns port0, THIS job's hub sits on a later candidate and may not even listen yet — so the spoke goes on to the
            # next candidate and raises only when `reject_grace` seconds after the first refusal (or the deadline) nobody has
            # taken it.
            deadline, last = time.time() + self.timeout, "no candidate port answered"
            refused_at, refused = None, None
            s = None
            while s is None:
                for port in self._candidates():
                    try:
                        c = socket.create_connection((self.addr, port), timeout=2.0)
                    except OSError as e:
                        last = "%s:%d %s" % (self.addr, port, e)
                        continue
                    try:
                        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        c.settimeout(3.0)
                        nonce = _recv_exact(c, 16)
                    except (OSError, ConnectionError) as e:      # silent or short: not a hub of ours, next candidate
                        c.close()
                        last = "%s:%d sent no challenge (%s: a foreign service?)" % (self.addr, port, type(e).__name__)
                        continue
                    # it speaks the protocol
                    mine, rk = os.urandom(16), struct.pack("<I", self.rank)
                    try:
                        c.settimeout(min(self.timeout, 10.0))
                        c.sendall(rk + mine + self._mac(b"spoke", nonce, rk))
                        proof = _recv_exact(c, 32)
                    except (OSError, ConnectionError):
                        c.close()
                        refused = "the hub on %s:%d rejected rank %d (token mismatch or duplicate rank)" % (self.addr, port, self.rank)
                        refused_at = refused_at or time.time()
                        continue
                    if not hmac.compare_digest(proof, self._mac(b"hub", mine)):
                        c.close()
                        refused = "%s:%d is not this job's hub (token mismatch)" % (self.addr, port)
                        refused_at = refused_at or time.time()
                        continue
                    c.settimeout(self.timeout)
                    s, self.port = c, port
                    break
                if s is None:
                    now = time.time()
                    if refused_at is not None and now > min(deadline, refused_at + self.reject_grace):
                        raise RuntimeError("rendezvous: a foreign hub refused rank %d and no hub of this job was found within %.0f s — %s"
                                           % (self.rank, self.reject_grace, refused))
                    if now > deadline:
                        raise RuntimeError("rendezvous: rank %d could not join the hub — %s" % (self.rank, last))
                    time.sleep(0.05)
            self._hub = s

    def all_gather(self, obj):
        """-> list of every rank's `obj` (None / int / float / bytes / ndarray / list of those), in rank order, on
        every rank."""
        if self.world == 1:
            return [obj]
        mine = _encode(obj)
        if self.rank == 0:
            parts = [mine] + [None] * (self.world - 1)
            for r, s in self._peers.items():
                parts[r] = _recv_msg(s)
            blob = b"".join(struct.pack("<Q", len(p)) + p for p in parts)
            for s in self._peers.values():
                _send_msg(s, blob)
        else:
            _send_msg(self._hub, mine)
            blob, parts, off = _recv_msg(self._hub), [], 0
            while off < len(blob):
                if off + 8 > len(blob):
                    raise ValueError("rendezvous: truncated gather frame")
                (ln,) = struct.unpack("<Q", blob[off:off + 8])
                if ln > len(blob) - off - 8:
                    raise ValueError("rendezvous: truncated gather frame")
                parts.append(blob[off + 8:off + 8 + ln])
                off += 8 + ln
            if len(parts) != self.world:
                raise ValueError("rendezvous: gather frame holds %d parts for %d ranks" % (len(parts), self.world))
        return [_decode(p) for p in parts]

    def broadcast(self, obj, root=0):
        return self.all_gather(obj if self.rank == root else None)[root]

    def barrier(self):
        self.all_gather(None)

    def max(self, value):
        return max(self.all_gather(float(value)))

    def close(self):
        for s in list(self._peers.values()) + [self._hub, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._hub, self._listener = {}, None, None


# ------------------------------------------------------------------------------------------------ RCCL side ----
COMM_ID_BYTES = 128


class PoseComm(object):
    """RCCL communicator bound to a `Context` (one per process/GPU)."""

    def __init__(self, ctx, rdzv):
        from .runtime import lib
        self.ctx, self.rdzv, self._lib = ctx, rdzv, lib
        self.rank, self.world = rdzv.rank, rdzv.world
        self._scalar = None
        if self.world > 1:
            # first-run de-risking (no N > 1 RCCL run exists): RCCL's own diagnostics go to stderr when — and only
            # when — something fails (WARN prints nothing on success), so the launcher's per-rank .err file explains a failed
            # ncclCommInitRank; the dmabuf IPC mode the pool's driver needs stays set for this process's children too
            os.environ.setdefault("NCCL_DEBUG", "WARN")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            uid = None
            if self.rank == 0:
                # a failure here (librccl missing, …) must still reach the broadcast, or every other rank waits for the id
                try:
                    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
                    lib.deepim_comm_unique_id(buf)
                    uid = buf.raw
                except Exception as e:      # noqa: BLE001 — shipped to the peers as text, re-raised below on every rank
                    uid = ("rank 0 could not make the RCCL id: %s" % e).encode()[:COMM_ID_BYTES - 1]
            uid = rdzv.broadcast(uid, 0)
            if not (isinstance(uid, bytes) and len(uid) == COMM_ID_BYTES):
                raise RuntimeError(uid.decode(errors="replace") if isinstance(uid, bytes) else "no RCCL id from rank 0")
            lib.deepim_comm_init(ctx.handle, self.rank, self.world, ctypes.create_string_buffer(uid, COMM_ID_BYTES))

    def all_gather_poses(self, all_poses, poses):
        """all_poses (world*B,3,4) ← poses (B,3,4) of every rank; device arrays; asynchronous on the context stream."""
        B = poses.shape[0]
        assert all_poses.shape[0] == self.world * B
        self._lib.deepim_allgather_poses(self.ctx.handle, all_poses, poses, B)

    def max_over_ranks(self, value):
        """Device all-reduce (MAX) of one float64 — bench.py's max-over-ranks step time."""
        if self.world == 1:
            return float(value)
        if self._scalar is None:
            self._scalar = self.ctx.empty((1,), dtype=np.float64)
        self._scalar.copyfrom(np.array([value], np.float64))
        self._lib.deepim_comm_allreduce_f64(self.ctx.handle, self._scalar, 1, 0)
        return float(self._scalar.asnumpy()[0])

    def close(self):
        if self.world > 1:
            self._lib.deepim_comm_destroy(self.ctx.handle)


def gather_padded(rdzv, local_poses, counts):
    """Host-side stand-in for the pose all-gather (CPU tests, ragged shards): numpy (B_local,3,4) of every rank →
    (sum B_local,3,4) in rank order.  On GPUs ragged shards pad to max(counts) around `deepim_allgather_poses`."""
    parts = rdzv.all_gather(np.ascontiguousarray(local_poses, np.float32))
    assert [len(p) for p in parts] == list(counts)
    return np.concatenate(parts, 0)
