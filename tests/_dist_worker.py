"""gloo worker (CPU, any world size) for tests/test_host.py::test_world_size_N_gloo_exchange: each rank reduces its shard
(oracle arithmetic) to ONE rank record, the PRODUCT's exchange_records makes all records known to all ranks, every rank merges
(device-free: the oracle's restatement of merge_kernel / merge_cov_kernel) -> identical to the unsharded update.
argv[1] = "plain" (516-float records: softmax partial + position sums) or "cov" (836 floats: + MPPI's second moments)."""
import sys

import numpy as np
import torch
import torch.distributed as dist

from covo_mpc_amd._lib import (COVO_COV_FLOATS, COVO_PARTIAL_FLOATS, COVO_POS_STATS_DOUBLES, COVO_RANK_RECORD_COV_FLOATS,
                               COVO_RANK_RECORD_FLOATS)
from covo_mpc_amd.controllers._core import exchange_records, shard_range
from oracle import ref_np as R
from oracle import rng_np


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "plain"
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N, lam, H = 1024, 0.01, 32
    off, n_local = shard_range(N, rank, world)
    # global-id keyed noise: the shard's rows equal rows [off, off+n_local) of the full draw
    eps = rng_np.randn(3, 4, off, n_local, 128)
    mu = (0.1 * np.sin(np.arange(128))).reshape(H, 4)
    a = np.clip(mu.reshape(-1)[None] + 0.5 * eps, -1, 1).astype(np.float64)
    cost = (np.abs(a).sum(axis=1) * 0.01).astype(np.float64)
    cov = kind == "cov"
    nrec = COVO_RANK_RECORD_COV_FLOATS if cov else COVO_RANK_RECORD_FLOATS
    n_part = COVO_PARTIAL_FLOATS + (COVO_COV_FLOATS if cov else 0)
    # the record in the product's layout (float32 words; the position sums ride as 192 doubles behind the partial): the
    # exchange moves bytes, so the fp64 test payload travels bit-exactly through a float64 VIEW of the same record
    rec = torch.zeros(nrec, dtype=torch.float64)  # one fp64 slot per product float: the oracle's arithmetic stays fp64
    iu = np.triu_indices(4)
    if cov:
        m, s, v, S2 = R.softmax_partial_cov(cost, a.reshape(n_local, H, 4), lam, mu)
        rec[COVO_PARTIAL_FLOATS:n_part] = torch.from_numpy(S2[:, iu[0], iu[1]].reshape(-1))  # 10 pairs i <= j per step
    else:
        m, s, v = R.softmax_partial(cost, a, lam)
    rec[0], rec[1] = float(m), float(s)
    rec[2:130] = torch.from_numpy(v)
    pos_sums = np.arange(COVO_POS_STATS_DOUBLES, dtype=np.float64) * (rank + 1)  # stand-in for the shard's position sums
    rec[n_part:n_part + COVO_POS_STATS_DOUBLES] = torch.from_numpy(pos_sums)
    gathered = torch.zeros((world * nrec,), dtype=torch.float64)
    g = exchange_records(rec, gathered).numpy()  # the ONE collective of a sharded control step (product host code)
    assert g.shape == (world, nrec)
    a_mean = mu
    eps_full = rng_np.randn(3, 4, 0, N, 128)
    a_full = np.clip(mu.reshape(-1)[None] + 0.5 * eps_full, -1, 1).astype(np.float64)
    cost_full = np.abs(a_full).sum(axis=1) * 0.01
    ref, w = R.softmax_update(cost_full, a_full.reshape(N, H, 4), lam, 0.9, a_mean)
    assert np.array_equal(eps, eps_full[off:off + n_local])
    if cov:
        S2s = np.zeros((world, H, 4, 4))
        S2s[:, :, iu[0], iu[1]] = g[:, COVO_PARTIAL_FLOATS:n_part].reshape(world, H, 10)
        S2s[:, :, iu[1], iu[0]] = S2s[:, :, iu[0], iu[1]]
        a_cov = np.tile(0.25 * np.eye(4), (H, 1, 1))
        merged, cov_new = R.merge_partials_cov(g[:, 0], g[:, 1], g[:, 2:130], S2s, lam, 0.9, a_mean, a_cov, 0.3)
        cov_ref = R.mppi_cov_update(w, a_full.reshape(N, H, 4), ref, a_cov, 0.3)
        assert np.abs(cov_new - cov_ref).max() < 1e-12, np.abs(cov_new - cov_ref).max()
    else:
        merged = R.merge_partials(g[:, 0], g[:, 1], g[:, 2:130], lam, 0.9, a_mean)
    assert np.abs(merged - ref).max() < 1e-12, np.abs(merged - ref).max()
    # the position sums of all ranks arrived in the same message
    tot = g[:, n_part:n_part + COVO_POS_STATS_DOUBLES].sum(axis=0)
    assert np.array_equal(tot, np.arange(COVO_POS_STATS_DOUBLES, dtype=np.float64) * (world * (world + 1) // 2))
    out = [None] * world
    dist.all_gather_object(out, merged.tobytes())
    assert all(o == out[0] for o in out)  # every rank merges identically
    dist.barrier()
    if rank == 0:
        print("DIST_OK", world, kind)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
