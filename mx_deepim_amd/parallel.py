"""Multi-GPU layout of the refinement loop, without PyTorch.

Pairs are independent units, sharded in contiguous blocks across ranks (one process per GPU); weights are
replicated; the only exchange on the data path is ONE all-gather of the refined (B_local,3,4) float32 poses per
refinement iteration — `deepim_allgather_poses` = `ncclAllGather` (RCCL over xGMI) enqueued on the library's own
stream, no host sync inside the loop.  The reference's counterpart is the per-device executor group merging the
outputs of all GPUs for the host (deepim/core/DataParallelExecutorGroup.py:364-388, deepim/test.py:135).

Host side:
  * `Rendezvous` — a tiny TCP star (rank 0 listens on MASTER_ADDR:MASTER_PORT+1, the others connect) used to ship
    the 128-byte RCCL unique id at start-up and for the few host-side collectives a launcher needs (barrier,
    all-gather of small Python objects, max-over-ranks of a timing).  It reads the same environment variables
    `python -m torch.distributed.run` sets (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT) but does not
    import torch.  Works without a GPU, which is how the world_size-2 CPU tests exercise it.
  * `PoseComm` — the RCCL communicator of a `Context`: `init` (bootstrap through a Rendezvous), `all_gather_poses`
    (device → device, asynchronous), `max_over_ranks` (device all-reduce of a float64).
  * `shard_bounds` / `shard_pairs` — the partition itself.
"""
import ctypes
import os
import pickle
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
    return _recv_exact(sock, n)


class Rendezvous(object):
    """TCP star between the ranks of one job: rank 0 is the hub.  Every collective is gather-to-hub + broadcast;
    payloads are small (ids, timings), this is control plane only."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0):
        env = os.environ
        self.rank = int(env.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(env.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.addr = addr or env.get("MASTER_ADDR", "127.0.0.1")
        # MASTER_PORT itself belongs to the launcher's own store; the job's rendezvous sits one above it
        self.port = int(env.get("DEEPIM_RDZV_PORT", int(env.get("MASTER_PORT", "29500")) + 1)) if port is None else int(port)
        self.timeout = float(timeout)
        self._peers = {}      # hub: rank -> socket
        self._hub = None      # spoke: socket to rank 0
        self._listener = None
        if self.world > 1:
            self._connect()

    def _connect(self):
        if self.rank == 0:
            ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            ls.bind((self.addr if self.addr not in ("localhost",) else "127.0.0.1", self.port))
            ls.listen(self.world)
            ls.settimeout(self.timeout)
            self._listener = ls
            while len(self._peers) < self.world - 1:
                conn, _ = ls.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(self.timeout)
                (r,) = struct.unpack("<I", _recv_exact(conn, 4))
                if r <= 0 or r >= self.world or r in self._peers:
                    conn.close()
                    raise RuntimeError("rendezvous: unexpected rank %d" % r)
                self._peers[r] = conn
        else:
            deadline = time.time() + self.timeout
            while True:
                try:
                    s = socket.create_connection((self.addr, self.port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise RuntimeError("rendezvous: rank %d could not reach %s:%d" % (self.rank, self.addr, self.port))
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(self.timeout)
            s.sendall(struct.pack("<I", self.rank))
            self._hub = s

    def all_gather(self, obj):
        """-> list of every rank's `obj` (picklable, small), in rank order, on every rank."""
        if self.world == 1:
            return [obj]
        mine = pickle.dumps(obj)
        if self.rank == 0:
            parts = [mine] + [None] * (self.world - 1)
            for r, s in self._peers.items():
                parts[r] = _recv_msg(s)
            blob = pickle.dumps(parts)
            for s in self._peers.values():
                _send_msg(s, blob)
        else:
            _send_msg(self._hub, mine)
            parts = pickle.loads(_recv_msg(self._hub))
        return [pickle.loads(p) for p in parts]

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
            uid = None
            if self.rank == 0:
                buf = ctypes.create_string_buffer(COMM_ID_BYTES)
                lib.deepim_comm_unique_id(buf)
                uid = buf.raw
            uid = rdzv.broadcast(uid, 0)
            assert isinstance(uid, bytes) and len(uid) == COMM_ID_BYTES
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
