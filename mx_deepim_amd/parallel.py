"""Multi-GPU layout of the refinement loop: pairs are independent units, sharded in
contiguous blocks across ranks (one process per GPU); the only exchange is an all-gather of the refined
(B_local,3,4) float32 poses per refinement iteration — RCCL over xGMI on GPUs ("nccl" backend), gloo in
the CPU tests.  No data-path collective besides that."""
import numpy as np


def shard_bounds(n_pairs, world_size, rank):
    """Contiguous block partition; the first (n_pairs % world_size) ranks take one extra pair."""
    base, extra = divmod(int(n_pairs), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_pairs(batch, world_size, rank, n_pairs=None):
    """Slice every per-pair array of a batch dict (leading axis = pairs; frame-major arrays have the pair
    axis second) to this rank's block."""
    n_pairs = n_pairs if n_pairs is not None else batch["image_observed"].shape[0]
    lo, hi = shard_bounds(n_pairs, world_size, rank)
    out = {}
    for k, v in batch.items():
        a = np.asarray(v)
        if a.ndim >= 1 and a.shape[0] == n_pairs:
            out[k] = a[lo:hi]
        elif a.ndim >= 2 and a.shape[1] == n_pairs:
            out[k] = a[:, lo:hi]
        else:
            out[k] = v
    return out


def all_gather_poses(local_poses, dist, counts=None):
    """local_poses: torch tensor (B_local, 3, 4) on this rank's device → (sum B_local, 3, 4) in rank order.
    `counts` (per-rank pair counts) is needed only for ragged shards."""
    import torch
    world = dist.get_world_size()
    flat = local_poses.reshape(local_poses.shape[0], 12).contiguous()
    if counts is None or len(set(counts)) == 1:
        out = torch.empty((world * flat.shape[0], 12), dtype=flat.dtype, device=flat.device)
        dist.all_gather_into_tensor(out, flat)
        return out.reshape(-1, 3, 4)
    mx = max(counts)
    pad = torch.zeros((mx, 12), dtype=flat.dtype, device=flat.device)
    pad[: flat.shape[0]] = flat
    out = torch.empty((world * mx, 12), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, pad)
    out = out.reshape(world, mx, 12)
    return torch.cat([out[r, : counts[r]] for r in range(world)], 0).reshape(-1, 3, 4)
