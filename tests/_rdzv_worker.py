"""Worker of tests/test_host_logic.py::test_rendezvous_under_torchrun_matches_gloo — launched by
`python -m torch.distributed.run`, exactly as the driver launches bench.py for N > 1.  Checks that the torch-free
Rendezvous reads the launcher's environment and that its gather agrees with torch.distributed's gloo all-gather
(torch is imported here, in a test, as the independent second opinion — never by the product)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mx_deepim_amd import parallel  # noqa: E402


def main():
    counts = [int(c) for c in sys.argv[1].split(",")]
    rd = parallel.Rendezvous()
    assert rd.world == len(counts) == int(os.environ["WORLD_SIZE"]) and rd.rank == int(os.environ["RANK"])
    allp = np.arange(sum(counts) * 12, dtype=np.float32).reshape(-1, 3, 4)
    lo, hi = parallel.shard_bounds(sum(counts), rd.world, rd.rank)
    assert hi - lo == counts[rd.rank] == parallel.shard_counts(sum(counts), rd.world)[rd.rank]
    got = parallel.gather_padded(rd, allp[lo:hi], counts)
    assert np.array_equal(got, allp)
    uid = rd.broadcast(bytes(range(128)) if rd.rank == 0 else None)
    assert uid == bytes(range(128))
    assert rd.max(float(rd.rank) + 0.5) == rd.world - 0.5
    rd.barrier()
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    mx = max(counts)
    pad = torch.zeros((mx, 12))
    pad[: hi - lo] = torch.from_numpy(allp[lo:hi].reshape(-1, 12))
    out = torch.empty((rd.world * mx, 12))
    dist.all_gather_into_tensor(out, pad)
    out = out.reshape(rd.world, mx, 12)
    ref = torch.cat([out[r, : counts[r]] for r in range(rd.world)], 0).reshape(-1, 3, 4).numpy()
    assert np.array_equal(ref, got)
    dist.destroy_process_group()
    rd.close()
    print("rank %d ok" % rd.rank)


if __name__ == "__main__":
    main()
