"""world_size-2 gloo worker for tests/test_host.py::test_world_size_2_gloo_exchange (CPU)."""
import numpy as np
import torch
import torch.distributed as dist

from covo_mpc_amd._lib import COVO_PARTIAL_FLOATS
from covo_mpc_amd.controllers._core import exchange_records, shard_range
from oracle import ref_np as R
from oracle import rng_np


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N, lam = 512, 0.01
    off, n_local = shard_range(N, rank, world)
    # global-id keyed noise: the shard's rows equal rows [off, off+n_local) of the full draw
    eps = rng_np.randn(3, 4, off, n_local, 128)
    a = np.clip(0.5 * eps, -1, 1).astype(np.float64)
    cost = (np.abs(a).sum(axis=1) * 0.01).astype(np.float64)
    m, s, v = R.softmax_partial(cost, a, lam)
    rec = torch.zeros(COVO_PARTIAL_FLOATS, dtype=torch.float64)
    rec[0], rec[1] = float(m), float(s)
    rec[2:130] = torch.from_numpy(v)
    gathered = torch.zeros((world * COVO_PARTIAL_FLOATS,), dtype=torch.float64)
    g = exchange_records(rec, gathered).numpy()  # the ONE collective of a sharded control step (product host code)
    a_mean = np.zeros((32, 4))
    merged = R.merge_partials(g[:, 0], g[:, 1], g[:, 2:130], lam, 1.0, a_mean)
    eps_full = rng_np.randn(3, 4, 0, N, 128)
    a_full = np.clip(0.5 * eps_full, -1, 1).astype(np.float64)
    cost_full = np.abs(a_full).sum(axis=1) * 0.01
    ref, _ = R.softmax_update(cost_full, a_full.reshape(N, 32, 4), lam, 1.0, a_mean)
    assert np.array_equal(eps, eps_full[off:off + n_local])
    assert np.abs(merged - ref).max() < 1e-12, np.abs(merged - ref).max()
    out = [None] * world
    dist.all_gather_object(out, merged.tobytes())
    assert out[0] == out[1]
    dist.barrier()
    if rank == 0:
        print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
