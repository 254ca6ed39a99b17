"""Shared device-side machinery of the sampling controllers (MPPI / CoVO).

Owns the libcovo_hip handle and the HBM buffers of one control step and issues the kernels of
include/covo_hip.h on torch's current stream.  When a torch.distributed process group with more
than one rank is given, the sample axis N is sharded: rank g owns global sample ids
[g*N/G, (g+1)*N/G), epsilon is keyed by global id, every rank reduces its shard to ONE rank record
(online-softmax partial + the position sums of covo.py:281 when they are wanted: 2 064 bytes) and ONE
exchange per control step makes all G records known to all ranks, which then merge identically
(SURVEY.md 5.8 / 8e).  The exchange is an all-gather (RCCL over xGMI under the nccl backend; the DEFAULT)
or -- opt-in: `exchange="peer"` / "auto", COVO_EXCHANGE=peer|auto -- direct peer writes into hipIpc-mapped buffers
(csrc/exchange.hip: no collective library and no host round trip, so whole sharded episodes can be enqueued from C).
The peer path has only ever run with ranks sharing ONE GPU (the build pool has single-GPU boxes): it stays opt-in
until it has been measured over xGMI.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .. import _lib
from .._lib import (COVO_COV_FLOATS, COVO_H, COVO_NA, COVO_PARTIAL_FLOATS, COVO_POS_STATS_DOUBLES, COVO_RANK_RECORD_COV_FLOATS,
                    COVO_RANK_RECORD_FLOATS, check, ptr)


def shard_range(N: int, rank: int, world: int):
    """Global sample ids owned by `rank`: [offset, offset + n_local)."""
    if N % world != 0:
        raise ValueError(f"N={N} must be divisible by the number of ranks {world}")
    n_local = N // world
    return rank * n_local, n_local


def exchange_records(record, gathered_flat, process_group=None):
    """THE one collective of a sample-sharded control step: all-gather the per-rank online-softmax
    records (RCCL over xGMI under the nccl backend; gloo in the CPU tests).  `gathered_flat` is a flat
    (world * len(record)) tensor; returns it viewed as (world, len(record))."""
    import torch.distributed as dist
    if record.is_cuda and dist.get_backend(process_group) == "gloo":
        # gloo has no device all-gather: stage the 528-byte records through the host (multi-process runs that share
        # one GPU, e.g. tests/test_gpu_parity.py::test_two_ranks_one_gpu; the product backend is nccl = RCCL)
        import torch
        g = torch.empty(gathered_flat.shape, dtype=gathered_flat.dtype)
        dist.all_gather_into_tensor(g, record.cpu(), group=process_group)
        gathered_flat.copy_(g)
    else:
        dist.all_gather_into_tensor(gathered_flat, record, group=process_group)
    return gathered_flat.view(-1, record.numel())


class SamplingCore:
    def __init__(self, N: int, H: int, lam: float, discount: float, device=None, process_group=None,
                 compute_info: bool = True, trust_clipped: bool = False, use_graph=None, shared_device=None, exchange=None,
                 cov_records: bool = False, propagate_nan=None):
        import torch
        if H != COVO_H:
            raise NotImplementedError(f"the fused kernels are built for H={COVO_H}, got H={H}")
        if not torch.cuda.is_available():
            raise _lib.CovoError("covo_mpc_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                                 "there is no CPU fallback")
        self.torch = torch
        self.lib = _lib.load_library()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (
            lambda i: torch.cuda.current_stream(i).cuda_stream)
        self.N, self.H, self.lam, self.discount = int(N), int(H), float(lam), float(discount)
        self.compute_info = compute_info
        self.pg = process_group
        self.world, self.rank = 1, 0
        if process_group is not None:
            import torch.distributed as dist
            self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        self.offset, self.n_local = shard_range(self.N, self.rank, self.world)
        # the action stripes this core rolls out always come from covo_noise_* (already clipped)
        # covo_mpc_step replays a captured hipGraph or issues its launches one by one (COVO_FLAG_NO_GRAPH).  Measured on the
        # MI355X box: every graph replay costs a ~8 us bubble on the GPU, an eager launch ~3.5 us of host
        # time; with one rank per node the host keeps ahead and eager is faster on every config (covo-online +3.4 %,
        # covo-offline / MPPI +18..25 %); a sample-sharded rank has less GPU work per step plus a collective to issue and is
        # better off with the graph's 40 us of host time.  COVO_GRAPH=1 / COVO_NO_GRAPH=1 force either.
        import os
        eager = self.world == 1
        if use_graph is not None:
            eager = not use_graph
        if os.environ.get("COVO_GRAPH") == "1":
            eager = False
        if os.environ.get("COVO_NO_GRAPH") == "1":
            eager = True
        self.uses_graph = not eager
        # shared_device (or COVO_SHARED_DEVICE=1): other processes / streams compete for this GPU, so no launch may rely on
        # its workgroups being co-resident -- the Sigma chain's two persistent launches (grid barriers) are replaced by one
        # launch per phase.  Without it a starved barrier times out after 0.2 s, that step's Sigma is NaN and the NEXT call
        # on the handle fails with COVO_E_DEVICE (covo_device_status).
        if shared_device is None:
            shared_device = os.environ.get("COVO_SHARED_DEVICE") == "1"
            if not shared_device and self.world > 1:
                # ranks of this group that sit on the SAME physical device compete for its CUs: detected here (one
                # object all-gather at construction), not left to the caller.  The identity is the PCI bus id, not the device
                # index: under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES every isolated rank sees its own GPU as index 0, and
                # comparing indices would put every real multi-GPU run on the slower one-launch-per-phase Sigma chain.
                import socket
                import torch.distributed as dist
                mine = (socket.gethostname(), self.device_bus_id())
                seen = [None] * self.world
                dist.all_gather_object(seen, mine, group=process_group)
                shared_device = seen.count(mine) > 1
        self.shared_device = bool(shared_device)
        # propagate_nan (or COVO_PROPAGATE_NAN=1): jnp.clip's NaN semantics in the action clips (COVO_FLAG_PROPAGATE_NAN) -- quadjax's
        # behaviour, where one NaN sample makes every later mean NaN; the default keeps the kernels' maxNum / minNum clip (NaN -> -1)
        if propagate_nan is None:
            propagate_nan = os.environ.get("COVO_PROPAGATE_NAN") == "1"
        self.propagate_nan = bool(propagate_nan)
        flags = ((_lib.COVO_FLAG_ACTIONS_CLIPPED if trust_clipped else 0) | (_lib.COVO_FLAG_NO_GRAPH if eager else 0) |
                 (_lib.COVO_FLAG_SHARED_DEVICE if self.shared_device else 0) |
                 (_lib.COVO_FLAG_PROPAGATE_NAN if self.propagate_nan else 0))
        cfg = _lib.ConfigC(self.n_local, self.H, 4, self.lam, self.discount, flags)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.covo_create(C.byref(cfg), C.byref(h)), "covo_create")
        self.h = h
        f32 = dict(dtype=torch.float32, device=self.device)
        n = self.n_local
        self._eps = None  # (n, 128) epsilon buffer, only materialised on request (parity tests)
        self._bufs = {}   # fixed-address buffers of the fused step
        self._args_cache = None
        self.a = torch.empty((COVO_H, n, 4), **f32)
        self.cost = torch.empty((n,), **f32)
        self.blockmin = torch.empty(((n + 63) // 64,), **f32)  # per-64-sample cost minima
        # the rank record of a sharded step: {m, s, v[128], pad[2]} + the 192 fp64 position sums, ONE message per step; on a
        # single rank the same buffer simply holds the two parts.  cov_records (MPPI with gamma_sigma != 0, mppi.py:119-125): the
        # record also carries the 320 weighted second moments between the two (COVO_RANK_RECORD_COV_FLOATS = 836 floats)
        self.cov_records = bool(cov_records)
        self.rec_floats = COVO_RANK_RECORD_COV_FLOATS if self.cov_records else COVO_RANK_RECORD_FLOATS
        n_part = COVO_PARTIAL_FLOATS + (COVO_COV_FLOATS if self.cov_records else 0)
        self.record = torch.zeros((self.rec_floats,), **f32)
        self.partial = self.record[:n_part]
        self.stats = self.record[n_part:].view(torch.float64)  # this shard's sums (528- / 1 808-byte offset: 8-aligned)
        self.gathered = torch.zeros((self.world * self.rec_floats,), **f32) if self.world > 1 else None
        self.stats_total = torch.zeros((COVO_POS_STATS_DOUBLES,), dtype=torch.float64, device=self.device) if self.world > 1 else self.stats
        self.exchange = "collective"
        if self.world > 1:
            # "collective" (default): torch.distributed's all-gather = RCCL over xGMI.  "peer": the peer-write exchange
            # (csrc/exchange.hip), raising if it cannot be set up.  "auto": peer when its construction-time self-test passes on EVERY
            # rank, else the collective.  Peer stays opt-in: it is validated on shared-GPU boxes only, never measured over xGMI.
            mode = exchange if exchange is not None else os.environ.get("COVO_EXCHANGE", "collective")
            if mode not in ("collective", "peer", "auto"):
                raise ValueError(f"exchange={mode!r} (auto | collective | peer)")
            if mode == "peer":
                self._connect_peer_exchange()
                self.exchange = "peer"
            elif mode == "auto":
                self.exchange = "peer" if self._try_peer_exchange() else "collective"

    def close(self):
        """covo_destroy now (workspaces, step graphs, side stream, exchange buffers) instead of at garbage collection.
        On sample-sharded ranks every rank closes at the same point: peers hold hipIpc mappings of the exchange buffer."""
        if getattr(self, "h", None):
            self.lib.covo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_bus_id(self) -> str:
        """PCI bus id of this core's GPU (covo_device_bus_id): the physical identity ranks compare, independent of
        HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES renumbering."""
        buf = C.create_string_buffer(32)
        check(self.lib.covo_device_bus_id(int(self._dev_index), buf, 32), "covo_device_bus_id")
        return buf.value.decode()

    def _connect_peer_exchange(self, must: bool = True) -> bool:
        """csrc/exchange.hip setup: every rank exports its exchange buffer (hipIpc), the 64-byte handles are all-gathered once
        over the process group, every rank maps all peers' buffers.  Every rank runs the SAME sequence of collectives whatever
        fails locally (a rank that raised between two of them would leave the others waiting in the next one); a failure
        anywhere makes every rank return False (must = False) or raise (must = True)."""
        import torch.distributed as dist
        hb = (C.c_char * _lib.COVO_EXCHANGE_HANDLE_BYTES)()
        err = None
        with self.torch.cuda.device(self.device):
            try:
                if os.environ.get("COVO_DEBUG_FAIL_EXCHANGE_RANK") == str(self.rank):  # tests: one rank cannot export its buffer
                    raise RuntimeError("covo_exchange_create: injected failure (COVO_DEBUG_FAIL_EXCHANGE_RANK)")
                check(self.lib.covo_exchange_create(self.h, self.world, self.rank, hb), "covo_exchange_create")
            except Exception as e:  # noqa: BLE001 - reported below, after the collective
                err = e
            handles = [None] * self.world
            dist.all_gather_object(handles, None if err else bytes(hb), group=self.pg)
            if all(x is not None for x in handles):
                try:
                    blob = (C.c_char * (self.world * _lib.COVO_EXCHANGE_HANDLE_BYTES)).from_buffer_copy(b"".join(handles))
                    check(self.lib.covo_exchange_connect(self.h, blob), "covo_exchange_connect")
                except Exception as e:  # noqa: BLE001
                    err = e
            elif err is None:
                err = RuntimeError("covo_exchange_create failed on another rank")
        # nobody pushes before everybody has mapped -- and everybody learns whether everybody has
        ok = self._all_agree(err is None)
        if not ok and must:
            raise err if err is not None else RuntimeError("covo_exchange_connect failed on another rank")
        return ok

    def _all_agree(self, ok: bool) -> bool:
        import torch.distributed as dist
        seen = [None] * self.world
        dist.all_gather_object(seen, bool(ok), group=self.pg)
        return all(seen)

    def _try_peer_exchange(self) -> bool:
        """exchange="auto": map the peers' buffers and run two exchanges (both parities) of a known record; the peer path is taken
        only if every rank mapped every buffer and read back every rank's record bit for bit.  A failure costs the wait kernel's
        bounded spin once (2 s for this probe; the product exchanges then wait COVO_EXCHANGE_TIMEOUT_S, default 60 s, like a
        collective that simply waits), leaves the handle clean (status cleared) and the collective in charge."""
        torch = self.torch
        # a rank that could not map must not leave the others spinning on its flag: all agree BEFORE anybody pushes
        if not self._connect_peer_exchange(must=False):
            return False
        ok = True
        keep = self.record.clone()
        try:
            check(self.lib.covo_exchange_set_timeout(self.h, 2.0), "covo_exchange_set_timeout")
            for rnd in range(2):
                probe = torch.arange(self.rec_floats, dtype=torch.float32, device=self.device) + 1000.0 * self.rank + 0.5 * rnd
                self.record.copy_(probe)
                self._peer_exchange()
                torch.cuda.synchronize(self.device)
                got = self.gathered.view(self.world, self.rec_floats).cpu()
                want = torch.stack([torch.arange(self.rec_floats, dtype=torch.float32) + 1000.0 * r + 0.5 * rnd
                                    for r in range(self.world)])
                ok = ok and bool(torch.equal(got, want))
            ok = ok and self.device_status() == 0
        except Exception:
            ok = False
        self.device_status(clear=True)
        self.record.copy_(keep)
        try:
            check(self.lib.covo_exchange_set_timeout(self.h, float(os.environ.get("COVO_EXCHANGE_TIMEOUT_S", "60"))),
                  "covo_exchange_set_timeout")
        except Exception:
            ok = False
        return self._all_agree(ok)

    def _peer_exchange(self):
        fn = self.lib.covo_exchange_records_cov if self.cov_records else self.lib.covo_exchange_records
        check(fn(self.h, ptr(self.record), ptr(self.gathered), self.stream()), "covo_exchange_records")

    def exchange_rank_records(self):
        """THE one exchange of a sharded control step: this rank's record -> self.gathered (world x rec_floats)."""
        if self.exchange == "peer":
            self._peer_exchange()
        else:
            exchange_records(self.record, self.gathered, self.pg)
        return self.gathered

    def merge_rank_records(self, a_mean_shifted, gamma_mean, out):
        # the stride / position-sum offset of the records THIS core exchanges: a core built with cov_records that runs a
        # gamma_sigma == 0 step still exchanges the 836-float kind ({m, s, v} in front, the sums at its end)
        check(self.lib.covo_merge_ranks_wide(self.h, ptr(self.gathered), self.world, int(self.rec_floats), ptr(a_mean_shifted),
                                             float(gamma_mean), ptr(out), ptr(self.stats_total) if self.compute_info else None,
                                             self.stream()), "covo_merge_ranks_wide")
        return out

    def merge_rank_records_cov(self, a_mean_shifted, gamma_mean, a_cov_shifted, gamma_sigma, out_mean, out_cov):
        """MPPI with gamma_sigma != 0 on sharded ranks (mppi.py:109-125): the gathered 836-float records -> new mean and the
        covariances adapted about it, identically on every rank (in place on a_cov_shifted is allowed)."""
        check(self.lib.covo_merge_ranks_cov(self.h, ptr(self.gathered), self.world, ptr(a_mean_shifted), float(gamma_mean),
                                            ptr(a_cov_shifted), float(gamma_sigma), ptr(out_mean), ptr(out_cov),
                                            ptr(self.stats_total) if self.compute_info else None, self.stream()),
              "covo_merge_ranks_cov")
        return out_mean, out_cov

    def device_status(self, clear: bool = False) -> int:
        """Sticky COVO_DEVSTAT_* bits raised by kernels of earlier calls (0 = fine); no synchronisation."""
        return int(self.lib.covo_device_status(self.h, 1 if clear else 0))

    # -- individual kernels -------------------------------------------------------------------
    def stream(self):
        # torch's current stream of this device, as the raw hipStream_t (the Stream-object route costs 4.5 us per call,
        # a sixth of the whole host path of a small-N step)
        return C.c_void_p(self._raw_stream(self._dev_index))

    def shift_mean(self, a_mean):
        out = self.torch.empty_like(a_mean)
        check(self.lib.covo_shift_mean(self.h, ptr(a_mean), ptr(out), self.stream()), "covo_shift_mean")
        return out

    @property
    def eps(self):
        if self._eps is None:
            self._eps = self.torch.empty((self.n_local, COVO_NA), dtype=self.torch.float32, device=self.device)
        return self._eps

    def randn(self, key):
        check(self.lib.covo_randn(self.h, int(key[0]), int(key[1]), self.offset, self.n_local, COVO_NA, ptr(self.eps),
                                  self.stream()), "covo_randn")
        return self.eps

    def randn_jax(self, act_key, mppi=False):
        """epsilon of this shard from jax.random's bitstream (random_jax.py / csrc/rng_jax.hip): what quadjax's
        split(act_key, N) + normal draws would be for the same key."""
        check(self.lib.covo_randn_jax(self.h, int(act_key[0]), int(act_key[1]), self.N, self.offset, self.n_local,
                                      1 if mppi else 0, ptr(self.eps), self.stream()), "covo_randn_jax")
        return self.eps

    def noise_gemm(self, L, mu, eps=None):
        eps = self.eps if eps is None else eps
        check(self.lib.covo_noise_gemm(self.h, ptr(L), ptr(mu), ptr(eps), self.n_local, ptr(self.a), self.stream()),
              "covo_noise_gemm")
        return self.a

    def noise_gemm_philox(self, L, mu, key):
        """covo_randn + covo_noise_gemm in one kernel: epsilon is drawn in registers (same values)."""
        check(self.lib.covo_noise_gemm_philox(self.h, ptr(L), ptr(mu), int(key[0]), int(key[1]), self.offset,
                                              self.n_local, ptr(self.a), self.stream()), "covo_noise_gemm_philox")
        return self.a

    def noise_blockdiag_philox(self, Ls, mu, key):
        check(self.lib.covo_noise_blockdiag_philox(self.h, ptr(Ls), ptr(mu), int(key[0]), int(key[1]), self.offset,
                                                   self.n_local, ptr(self.a), self.stream()),
              "covo_noise_blockdiag_philox")
        return self.a

    def noise_blockdiag(self, Ls, mu, eps=None):
        eps = self.eps if eps is None else eps
        check(self.lib.covo_noise_blockdiag(self.h, ptr(Ls), ptr(mu), ptr(eps), self.n_local, ptr(self.a),
                                            self.stream()), "covo_noise_blockdiag")
        return self.a

    def cholesky(self, A, n, batch):
        out = self.torch.empty_like(A)
        check(self.lib.covo_cholesky(self.h, ptr(A), n, batch, ptr(out), self.stream()), "covo_cholesky")
        return out

    def disturb_table(self, params_c, packed, key=None, keys_dev=None, key_mode=_lib.DISTURB_KEYS_SHARED, deterministic=True,
                      batch=1):
        """covo_disturb_table: the per-step disturbance table(s) [batch, H, 4] of rollouts starting at `packed` ([batch, 32]
        device states) for params_c.disturb_kind periodic / sin / drag / mixed (free.py:10-58).  `key` (two uint32) for every
        entry, or `keys_dev` (uint32 [batch, 2] device tensor): the key whose threading `key_mode` describes."""
        out = self.torch.empty((batch, COVO_H, 4), dtype=self.torch.float32, device=self.device)
        k0, k1 = (int(key[0]), int(key[1])) if key is not None else (0, 0)
        check(self.lib.covo_disturb_table(self.h, C.byref(params_c), ptr(packed), int(batch), ptr(keys_dev), k0, k1,
                                          int(key_mode), 1 if deterministic else 0, ptr(out), self.stream()), "covo_disturb_table")
        return out

    def rollout(self, dstate, params_c, f_shared, want_stats, f_steps=None):
        """f_shared: host 3-vector (none / gaussian); f_steps: the device table of disturb_table (periodic / sin / drag / mixed)."""
        fs = (C.c_float * 3)(*[float(x) for x in f_shared])
        check(self.lib.covo_rollout_cost(self.h, ptr(dstate.packed), ptr(dstate.pos_traj), ptr(dstate.vel_traj),
                                         dstate.T, C.byref(params_c), fs, ptr(f_steps), ptr(self.a), self.n_local, ptr(self.cost),
                                         ptr(self.blockmin), ptr(self.stats) if want_stats else None, self.stream()),
              "covo_rollout_cost")
        return self.cost

    def time_rollout(self, dstate, params_c, f_shared=(0.0, 0.0, 0.0), reps=100, with_records=False, f_steps=None):
        """(mean, min-batch) GPU microseconds per covo_rollout_cost launch: three batches of `reps` launches issued back to
        back from C between two events.  with_records: the record-emitting variant the fused step runs."""
        fs = (C.c_float * 3)(*[float(x) for x in f_shared])
        us = (C.c_float * 2)(0.0, 0.0)
        check(self.lib.covo_debug_time_rollout(self.h, ptr(dstate.packed), ptr(dstate.pos_traj), ptr(dstate.vel_traj), dstate.T,
                                               C.byref(params_c), fs, ptr(f_steps), ptr(self.a), self.n_local, ptr(self.cost),
                                               ptr(self.blockmin), 1 if with_records else 0, int(reps), us, self.stream()),
              "covo_debug_time_rollout")
        return float(us[0]), float(us[1])

    def hessian(self, packed, dstate, params_c, a_mean, batch=1, method="adjoint", f_steps=None):
        """d^2 C / da^2 (covo.py:134-185); method "adjoint" (default, hessian_adj.hip) or "pairs" (hessian.hip).  f_steps: the
        [batch, H, 4] table of disturb_table(key_mode=DISTURB_KEYS_HESSIAN) for periodic / sin / drag / mixed."""
        R = self.torch.empty((batch, COVO_NA, COVO_NA), dtype=self.torch.float64, device=self.device)
        fn = self.lib.covo_hessian if method == "adjoint" else self.lib.covo_hessian_pairs
        check(fn(self.h, ptr(packed), ptr(dstate.pos_traj), ptr(dstate.vel_traj), dstate.T,
                 C.byref(params_c), ptr(a_mean), ptr(f_steps), batch, ptr(R), self.stream()), "covo_hessian")
        return R

    def sigma(self, R, sample_sigma, batch=1, method="ns"):
        """(Sigma, chol(Sigma)); method "ns" = eigh-free GEMM pipeline (default), "jacobi" = eigendecomposition."""
        f32 = dict(dtype=self.torch.float32, device=self.device)
        Sigma = self.torch.empty((batch, COVO_NA, COVO_NA), **f32)
        L = self.torch.empty((batch, COVO_NA, COVO_NA), **f32)
        fn = self.lib.covo_sigma if method == "ns" else self.lib.covo_sigma_jacobi
        check(fn(self.h, ptr(R), batch, float(sample_sigma), ptr(Sigma), ptr(L), self.stream()), "covo_sigma")
        return Sigma, L

    # -- the whole step in one C call (hipGraph-replayed from its third invocation) ---------------
    def _persistent(self, name, shape, dtype=None):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self.torch.zeros(shape, dtype=dtype or self.torch.float32, device=self.device)
            self._bufs[name] = t
        return t

    def _prepare_step(self, mode, dstate, a_mean, *, a_cov=None, L_table=None, gamma_mean=1.0, sample_sigma=0.5,
                      want_stats=False, derive_keys=False, rollout_deterministic=True, gamma_sigma=0.0, carry_only=False):
        """Fixed-address buffers + struct covo_step_args of the fused step -> (args, a_mean buffer, shifted-mean
        buffer, a_cov buffer or None)."""
        torch = self.torch
        # the state is read through a pointer that travels with the per-step scalars (step.hip): no copy, and a
        # new address does not invalidate the captured graph.  The tensor must stay alive until the step has run.
        packed = dstate.packed
        self._state_ref = packed
        am = self._persistent("a_mean", (COVO_NA,))
        # control_params.a_mean is an INPUT of the call (covo.py:201): when it is not the handle's own buffer (a returned view of
        # it) the step's first launch reads it where it lies (args.a_mean_in) -- no copy launch on the per-step path
        a_mean_in = None
        if a_mean.data_ptr() != am.data_ptr():
            if (not carry_only and a_mean.is_cuda and a_mean.device == am.device and a_mean.dtype == torch.float32 and
                    a_mean.is_contiguous() and a_mean.numel() == COVO_NA):  # (another GPU's tensor is copied: ADVICE r05)
                a_mean_in = a_mean
                cur = torch.cuda.current_stream(am.device)
                if cur.cuda_stream != 0:  # the tensor may have been allocated on another stream: keep its block until this one is through
                    a_mean.record_stream(cur)
            else:
                am.copy_(a_mean.reshape(-1), non_blocking=True)
        am_shift = self._persistent("a_mean_shift", (COVO_NA,))
        cov_out = None
        if mode == _lib.MODE_COVO_ONLINE:
            cov_out = self._persistent("a_cov", (COVO_NA, COVO_NA))
        elif mode == _lib.MODE_MPPI:
            cov_out = self._persistent("a_cov_mppi", (COVO_H, 4, 4))
            if a_cov.data_ptr() != cov_out.data_ptr():
                cov_out.copy_(a_cov, non_blocking=True)
        # the argument block only changes when a buffer does (new episode -> new trajectory tensors): it is rebuilt
        # then, otherwise only the state pointer is refreshed (this call sits on the per-step host path)
        sig = (mode, dstate.pos_traj.data_ptr(), dstate.vel_traj.data_ptr(), L_table.data_ptr() if L_table is not None else 0,
               bool(want_stats), float(gamma_mean), float(sample_sigma), bool(derive_keys), bool(rollout_deterministic),
               float(gamma_sigma))
        cached = self._args_cache
        if cached is not None and cached[0] == sig:
            args = cached[1]
        else:
            assert packed.is_contiguous() and packed.dtype == torch.float32 and packed.numel() == _lib.COVO_STATE_FLOATS
            args = _lib.StepArgsC()
            args.mode, args.n_samples, args.T = mode, self.n_local, dstate.T
            args.pos_traj, args.vel_traj = dstate.pos_traj.data_ptr(), dstate.vel_traj.data_ptr()
            args.a_mean, args.a_mean_shift = am.data_ptr(), am_shift.data_ptr()
            if cov_out is not None:
                args.a_cov = cov_out.data_ptr()
            if mode == _lib.MODE_COVO_OFFLINE:
                args.L_table, args.n_table = L_table.data_ptr(), int(L_table.shape[0])
            args.a, args.cost, args.groupmin = self.a.data_ptr(), self.cost.data_ptr(), self.blockmin.data_ptr()
            args.pos_stats = self.stats.data_ptr() if want_stats else None
            args.partial_out = self.partial.data_ptr() if self.world > 1 else None
            args.sample_offset, args.gamma_mean, args.sample_sigma = self.offset, float(gamma_mean), float(sample_sigma)
            args.derive_keys, args.rollout_deterministic = (1 if derive_keys else 0), (1 if rollout_deterministic else 0)
            args.gamma_sigma = float(gamma_sigma)
            self._args_cache = (sig, args, (dstate.pos_traj, dstate.vel_traj, L_table))  # keep the tensors alive
        args.state = packed.data_ptr()
        args.a_mean_in = a_mean_in.data_ptr() if a_mean_in is not None else None
        self._a_mean_in_ref = a_mean_in  # alive until the step has run
        return args, am, am_shift, cov_out

    def step(self, mode, dstate, params_c, a_mean, key, *, f_shared=None, gamma_mean=1.0, **kw):
        """covo_mpc_step: returns (a_mean_new, a_cov_out) as views of persistent buffers (clone to keep).
        derive_keys: `key` is the controller's raw rng_act; the sampling key and MPPI's shared disturbance are derived
        from it on the device (step.hip) -- the host does no RNG work on the per-step path."""
        args, am, am_shift, cov_out = self._prepare_step(mode, dstate, a_mean, gamma_mean=gamma_mean, **kw)
        fs = (C.c_float * 3)(*[float(x) for x in f_shared]) if f_shared is not None else None
        check(self.lib.covo_mpc_step(self.h, C.byref(params_c), C.byref(args), int(key[0]), int(key[1]), fs, self.stream()),
              "covo_mpc_step")
        self._last_step = (params_c, args)
        if self.world > 1:
            self.exchange_rank_records()  # the ONE exchange per step (partial (+ second moments) + position sums in one record)
            gs = float(kw.get("gamma_sigma", 0.0))
            if mode == _lib.MODE_MPPI and gs != 0.0:
                self._need_cov_records()
                self.merge_rank_records_cov(am_shift, gamma_mean, cov_out, gs, am, cov_out)  # a_cov was shifted by the begin launch
            else:
                self.merge_rank_records(am_shift, gamma_mean, am)
        return am, cov_out

    def run_episode(self, mode, episode, params_c, a_mean, rng, n_steps, **kw):
        """covo_run_episode: n_steps x (fused control step on episode.noisy -> env step on the device) enqueued by ONE
        C call, keys threaded like eval_env's run_one_step (quadrotor.py:520-538).  -> (a_mean buffer, a_cov buffer,
        rng after the segment).  Asynchronous; episode.read_log() synchronises."""
        if self.world > 1 and self.exchange != "peer":
            raise NotImplementedError("run_episode on sample-sharded ranks needs the peer-write exchange (exchange='peer' / "
                                      "COVO_EXCHANGE=peer): a torch.distributed collective cannot be enqueued from C")
        if self.world > 1 and mode == _lib.MODE_MPPI and float(kw.get("gamma_sigma", 0.0)) != 0.0:
            self._need_cov_records()
        args, am, _, cov_out = self._prepare_step(mode, episode.noisy_state, a_mean, derive_keys=True, carry_only=True, **kw)
        key = (C.c_uint32 * 2)(int(rng[0]), int(rng[1]))
        env = episode.env
        # the env step's auto-reset (base.py:22-40) is a property of the EPISODE, the model constants come from the controller
        params_c = type(params_c).from_buffer_copy(params_c)
        for f in ("reset_traj", "reset_dt", "reset_disturb_scale"):
            setattr(params_c, f, getattr(episode.params_c, f))
        check(self.lib.covo_run_episode(self.h, C.byref(params_c), C.byref(args), ptr(episode.true), ptr(episode.acc_traj),
                                        1 if env.generate_noisy_state else 0, float(env.default_params.obs_noise_scale),
                                        ptr(episode.log[episode.n_steps:]), key, int(n_steps), self.stream()),
              "covo_run_episode")
        episode.n_steps += int(n_steps)
        self._last_step = (params_c, args)
        import numpy as np
        return am, cov_out, np.array([key[0], key[1]], dtype=np.uint32)

    def time_phases(self, step_mask=63, hess_mask=15, sigma_stages=4, reps=20):
        """GPU microseconds of the selected launches of the LAST step() call, replayed `reps` times from one graph."""
        params_c, args = self._last_step
        us = C.c_float(0.0)
        self.torch.cuda.synchronize()
        check(self.lib.covo_debug_time_step(self.h, C.byref(params_c), C.byref(args), step_mask, hess_mask, sigma_stages,
                                            reps, C.byref(us), self.stream()), "covo_debug_time_step")
        return float(us.value)

    def update(self, a_mean_shifted, gamma_mean):
        """softmax weights + weighted mean (+ the one collective when sharded) -> new mean (H,4)."""
        out = self.torch.empty_like(a_mean_shifted)
        if self.world == 1:
            check(self.lib.covo_softmax_update(self.h, ptr(self.cost), ptr(self.a), self.n_local, ptr(self.blockmin),
                                               ptr(a_mean_shifted), float(gamma_mean), ptr(out), self.stream()),
                  "covo_softmax_update")
            return out
        check(self.lib.covo_softmax_reduce(self.h, ptr(self.cost), ptr(self.a), self.n_local, ptr(self.blockmin),
                                           ptr(self.partial), self.stream()), "covo_softmax_reduce")
        self.exchange_rank_records()  # the ONE exchange per step
        return self.merge_rank_records(a_mean_shifted, gamma_mean, out)

    def _need_cov_records(self):
        if not self.cov_records:
            raise NotImplementedError("MPPI covariance adaptation (gamma_sigma != 0) on sample-sharded ranks needs the 836-float rank "
                                      "records: build the controller with control_params.gamma_sigma != 0 "
                                      "(SamplingCore(cov_records=True))")

    def update_cov(self, a_mean_shifted, gamma_mean, a_cov_shifted, gamma_sigma):
        """MPPI's update with covariance adaptation (mppi.py:109-125) -> (new mean (128,), new a_cov (H,4,4)).  Sharded ranks:
        this shard's record with its second moments -> the ONE exchange -> the same merge on every rank."""
        mean = self.torch.empty_like(a_mean_shifted)
        cov = self.torch.empty_like(a_cov_shifted)
        if self.world > 1:
            self._need_cov_records()
            check(self.lib.covo_softmax_reduce_cov(self.h, ptr(self.cost), ptr(self.a), self.n_local, ptr(self.blockmin),
                                                   ptr(a_mean_shifted), ptr(self.partial), self.stream()), "covo_softmax_reduce_cov")
            self.exchange_rank_records()
            return self.merge_rank_records_cov(a_mean_shifted, gamma_mean, a_cov_shifted, gamma_sigma, mean, cov)
        check(self.lib.covo_softmax_update_cov(self.h, ptr(self.cost), ptr(self.a), self.n_local, ptr(self.blockmin),
                                               ptr(a_mean_shifted), float(gamma_mean), ptr(a_cov_shifted), float(gamma_sigma),
                                               ptr(mean), ptr(cov), self.stream()), "covo_softmax_update_cov")
        return mean, cov

    def info(self, dstate):
        """{"pos_mean","pos_std"} (H,3) from the per-step sums (controllers/covo.py:281): covo_pos_info, one launch."""
        torch = self.torch
        stats = self.stats_total  # sharded: the ranks' sums arrived in the rank records and were added by covo_merge_ranks
        out = torch.empty((2, COVO_H, 3), dtype=torch.float32, device=self.device)
        check(self.lib.covo_pos_info(self.h, ptr(stats), ptr(dstate.packed), int(self.N), ptr(out[0]), ptr(out[1]), self.stream()),
              "covo_pos_info")
        return {"pos_mean": out[0], "pos_std": out[1]}
