"""CoVO-MPC controller behind quadjax's call signature: quadjax/controllers/covo.py:13-283.

Differs from MPPI only in how the sampling covariance is chosen (covo.py:205-208): `online`
takes the exact Hessian of the rollout cost at the shifted mean (second-order adjoint on fp64
MFMAs, csrc/hessian_adj.hip; the per-pair hyper-dual kernel of csrc/hessian.hip is the
independent cross-check), forms the optimal Sigma = c (R + delta I)^(-1/2) WITHOUT an
eigendecomposition (Chebyshev-filter lambda_min + Rayleigh-Ritz, coupled Newton-Schulz, one
Cholesky: csrc/sigma_ns.hip; the Jacobi eigensolver of csrc/sigma.hip is the cross-check) and
samples from its factor -- one C call per control step (covo_mpc_step, csrc/step.hip);
`offline` looks Sigma up in a per-episode table built at reset() along a PID-tracked nominal
trajectory (covo.py:44-112; csrc/pid_nominal.hip + the batched Hessian / Sigma launches).
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import Any, Optional

import numpy as np

from .._lib import COVO_NA
from ..dynamics.dataclass import as_device_state
from .base import BaseController
from .pid import PIDController, PIDParams
from ._core import SamplingCore


@dataclass(frozen=True)
class CoVOParams:
    """covo.py:13-22.  a_mean (H,4), a_cov (128,128), a_cov_offline (T,128,128): torch fp32 on the GPU.
    a_chol_offline caches the table's Cholesky factors (the reference re-factors every step inside
    multivariate_normal, covo.py:216; the result is identical)."""
    gamma_mean: float
    gamma_sigma: float
    discount: float
    sample_sigma: float
    a_mean: Any
    a_cov: Any
    a_cov_offline: Any
    a_chol_offline: Optional[Any] = None

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


class CoVOController(BaseController):
    def __init__(self, env, control_params, N: int, H: int, lam: float, mode: str = "online", *, device=None,
                 process_group=None, compute_info: bool = True, propagate_nan=None) -> None:
        super().__init__(env, control_params)
        self.N, self.H, self.lam = N, H, lam
        self.materialize_eps = False  # True: epsilon is written to HBM and the kernels are called one by one (parity)
        self.noise_stream = "philox"  # "jax": keys split and epsilon drawn from jax.random's own bitstream (random_jax.py,
                                      # csrc/rng_jax.hip; kernel-by-kernel path) -- replayable on a JAX-equipped machine
        self.alias_outputs = False    # True: returned a_mean / a_cov alias the controller's buffers (no clones)
        self.action_dim = self.env.action_dim
        if mode not in ("online", "offline"):
            raise NotImplementedError(mode)  # covo.py:113-114
        self._params_c(env.default_params)  # raises now if env.reward_fn / disturb_type is not one the kernels evaluate
        if mode == "offline":
            assert env.action_dim == 4, "only support 4D action space Quadrotor environment for now"  # covo.py:45-47
            self.expansion_control_params = PIDParams(Kp=10.0, Kd=5.0, Ki=0.0, Kp_att=10.0)  # covo.py:48-53
            self.expansion_controller = PIDController(env, control_params=control_params)
            self.reset = self.reset_a_cov_offline  # covo.py:112
        self.mode = mode
        # propagate_nan: jnp.clip's NaN semantics in the sampling clip (covo.py:224) -- see SamplingCore
        self.core = SamplingCore(N, H, lam, control_params.discount, device=device, process_group=process_group,
                                 compute_info=compute_info, trust_clipped=True, propagate_nan=propagate_nan)

    def _needs_table(self, params_c) -> bool:
        """periodic / sin / drag / mixed (free.py:10-58): quadjax's deterministic=True only zeroes dyn_noise_scale
        (quadrotor.py:234-235), so these models act in the sampling rollouts AND in get_hessian -- through the per-step
        table of csrc/disturb.hip."""
        from .. import _lib
        return params_c.disturb_kind in _lib.TABLE_DISTURB_KINDS

    # ---- Sigma selection (covo.py:36-41 / 107-108) ----------------------------------------------
    def get_hessian(self, env_state, env_params, control_params, a_mean, rng_act=None):
        """covo.py:134-185 -> (128,128) fp64 tensor.  rng_act: the key whose per-step splits (covo.py:150-153) feed the
        disturbance model's draws (periodic / mixed)."""
        from .. import _lib
        dstate = as_device_state(env_state, self.core.device)
        pc = self._params_c(env_params)
        tab = None
        if self._needs_table(pc):
            if rng_act is None:
                raise ValueError(f"disturb_type={self.env.disturb_type!r}: get_hessian needs rng_act")
            tab = self.core.disturb_table(pc, dstate.packed, key=rng_act, key_mode=_lib.DISTURB_KEYS_HESSIAN, deterministic=True)
        return self.core.hessian(dstate.packed, dstate, pc, a_mean.reshape(-1), f_steps=tab)[0]

    def optimize_sigma(self, R, control_params):
        """covo.py:116-132 -> (Sigma, chol(Sigma)) fp32."""
        Sigma, L = self.core.sigma(R.reshape(1, COVO_NA, COVO_NA), control_params.sample_sigma)
        return Sigma[0], L[0]

    # ---- covo-offline reset (covo.py:58-104) --------------------------------------------------------
    def _nominal_host(self, env_state, env_params, key):
        """The nominal states / means of covo.py:58-99 by the Python env and PID (300 x 33 scalar steps, ~1.5 s):
        the restatement the device kernels (csrc/pid_nominal.hip) are tested against."""
        from .. import random as crandom
        env = self.env
        T = env.default_params.max_steps_in_episode
        packed = np.zeros((T, 32), dtype=np.float32)
        a_means = np.zeros((T, COVO_NA), dtype=np.float32)
        s = env_state
        for t in range(T):  # get_single_a_cov_offline, covo.py:72-90
            # nominal mean: H deterministic PID steps (covo.py:58-76)
            sr, kr = s, key
            for k in range(self.H):
                rng_act, kr = crandom.split(kr)
                action, _, _ = self.expansion_controller(None, sr, env_params, rng_act, self.expansion_control_params)
                rng_step, kr = crandom.split(kr)
                _, sr, _, _, _ = env.step_env(rng_step, sr, action, env_params, deterministic=True, need_info=False)
                a_means[t, 4 * k:4 * k + 4] = action
            packed[t] = s.pack()
            # advance the real state one PID step, deterministic=False (covo.py:80-90)
            rng_step, key = crandom.split(key)
            action, _, _ = self.expansion_controller(None, s, env_params, rng_step, self.expansion_control_params)
            rng_step, key = crandom.split(key)
            _, s, _, _, _ = env.step_env(rng_step, s, action, env_params, need_info=False)
        return packed, a_means

    def _nominal_device(self, env_state, env_params, key):
        """The same on the device: one launch walks the PID-tracked chain of start states, one rolls the H nominal steps
        of every start state (covo_pid_nominal)."""
        import ctypes as C
        from .. import _lib
        core, env = self.core, self.env
        torch = core.torch
        T = env.default_params.max_steps_in_episode
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(core.device)
        dstate = as_device_state(env_state, core.device)
        state0, acc_traj = up(env_state.pack()), up(env_state.acc_traj)
        packed_d = torch.empty((T, 32), dtype=torch.float32, device=core.device)
        a_means_d = torch.empty((T, COVO_NA), dtype=torch.float32, device=core.device)
        keys_d = torch.empty((T, 2), dtype=torch.int32, device=core.device)  # uint32 bits: the scan's carry key per start state
        g = self.expansion_control_params
        assert float(g.Ki) == 0.0, "the device PID law carries no integral state (covo.py:48-53 uses Ki = 0)"
        pc, pid_pc = self._params_c(env_params), self.expansion_controller.param.to_c()
        _lib.check(core.lib.covo_pid_nominal(core.h, _lib.ptr(state0), _lib.ptr(dstate.pos_traj), _lib.ptr(dstate.vel_traj),
                                             _lib.ptr(acc_traj), dstate.T, C.byref(pc), C.byref(pid_pc), float(g.Kp),
                                             float(g.Kd), float(g.Kp_att), int(key[0]), int(key[1]), T,
                                             _lib.ptr(packed_d), _lib.ptr(a_means_d), _lib.ptr(keys_d), core.stream()),
                   "covo_pid_nominal")
        self._nominal_keep = (state0, acc_traj)  # alive until the stream has consumed them
        return packed_d, a_means_d, keys_d

    def reset_a_cov_offline(self, env_state, env_params, control_params, key):
        from .. import _lib
        core = self.core
        T = self.env.default_params.max_steps_in_episode
        dstate = as_device_state(env_state, core.device)
        packed_d, a_means_d, keys_d = self._nominal_device(env_state, env_params, key)
        pc = self._params_c(env_params)
        tab = None
        if self._needs_table(pc):  # get_hessian(env_state_t, ..., a_mean_t, key_t) (covo.py:77): per-step keys from the carry key
            tab = core.disturb_table(pc, packed_d, keys_dev=keys_d, key_mode=_lib.DISTURB_KEYS_HESSIAN, deterministic=True, batch=T)
        R = core.hessian(packed_d, dstate, pc, a_means_d, batch=T, f_steps=tab)
        Sigma, L = core.sigma(R, control_params.sample_sigma, batch=T)
        return control_params.replace(a_cov_offline=Sigma, a_chol_offline=L)

    def run_episode(self, episode, env_params, control_params, rng, n_steps):
        """n_steps closed-loop steps (this controller + the device env step) enqueued by one C call; keys threaded like
        eval_env's run_one_step.  -> (control_params with the final mean / Sigma, rng).  See SamplingCore.run_episode."""
        from .. import _lib
        if self.mode == "offline" and control_params.a_chol_offline is None:
            raise RuntimeError("covo-offline: call controller.reset(...) first (a_cov_offline table missing)")
        mode = _lib.MODE_COVO_ONLINE if self.mode == "online" else _lib.MODE_COVO_OFFLINE
        am, cov, rng = self.core.run_episode(mode, episode, self._params_c(env_params), control_params.a_mean, rng, n_steps,
                                             L_table=control_params.a_chol_offline, gamma_mean=control_params.gamma_mean,
                                             sample_sigma=control_params.sample_sigma)
        a_mean = am.view(self.H, 4) if self.alias_outputs else am.view(self.H, 4).clone()
        if self.mode == "online":
            control_params = control_params.replace(a_mean=a_mean, a_cov=cov if self.alias_outputs else cov.clone())
        else:
            control_params = control_params.replace(a_mean=a_mean)
        return control_params, rng

    # ---- one MPC control step (covo.py:187-283) -----------------------------------------------------
    def __call__(self, obs, env_state, env_params, rng_act, control_params: CoVOParams, info):
        from .. import random as crandom
        from .. import _lib
        core = self.core
        dstate = as_device_state(info["noisy_state"], core.device)  # covo.py:198
        params_c = self._params_c(env_params)
        if self.mode == "offline" and control_params.a_chol_offline is None:
            raise RuntimeError("covo-offline: call controller.reset(...) first (a_cov_offline table missing)")
        if self.noise_stream not in ("philox", "jax"):
            raise ValueError(f"noise_stream={self.noise_stream!r}")
        if not self.materialize_eps and self.noise_stream == "philox":
            # ---- production path: the whole step is one C call / one hipGraph replay (csrc/step.hip)
            # rng_act, act_key = split(rng_act) (covo.py:212), the rollouts' step_key (covo.py:225: deterministic, so only the
            # periodic / mixed models draw from it) and get_hessian's per-step keys (covo.py:39,150-153) are derived on the
            # device from the raw key (step.hip: step_begin_kernel, disturb.hip)
            mode = _lib.MODE_COVO_ONLINE if self.mode == "online" else _lib.MODE_COVO_OFFLINE
            am, cov = core.step(mode, dstate, params_c, control_params.a_mean, rng_act,
                                L_table=control_params.a_chol_offline, gamma_mean=control_params.gamma_mean,
                                sample_sigma=control_params.sample_sigma, want_stats=core.compute_info,
                                derive_keys=True)
            a_mean_new = am.view(self.H, 4)
            if self.mode == "online":
                a_cov = cov
            elif dstate.time is not None:  # covo.py:107-108, JAX gather clamps
                a_cov = control_params.a_cov_offline[min(max(dstate.time, 0), control_params.a_cov_offline.shape[0] - 1)]
            else:
                t_idx = dstate.packed[25:26].view(core.torch.int32).long().clamp(0, control_params.a_cov_offline.shape[0] - 1)
                a_cov = control_params.a_cov_offline.index_select(0, t_idx)[0]
            if not self.alias_outputs:
                a_mean_new = a_mean_new.clone()
                a_cov = a_cov.clone() if self.mode == "online" else a_cov
            control_params = control_params.replace(a_mean=a_mean_new, a_cov=a_cov)
            out_info = core.info(dstate) if core.compute_info else {}
            return a_mean_new[0], control_params, out_info

        # ---- kernel-by-kernel path with epsilon materialised in HBM (identical values; parity/debug)
        a_mean = core.shift_mean(control_params.a_mean.reshape(-1))  # covo.py:201-203
        control_params = control_params.replace(a_mean=a_mean.view(self.H, 4))
        tables = self._needs_table(params_c)
        jax_stream = self.noise_stream == "jax"

        def table(key, key_mode):
            # Philox stream: built on the device (disturb.hip); jax stream: the same table on the host with jax.random's own key
            # splits and uniform draws (envs/quadrotor.py: rollout_disturbance_table), uploaded (128 floats)
            if not tables:
                return None
            if not jax_stream:
                return core.disturb_table(params_c, dstate.packed, key=key, key_mode=key_mode, deterministic=True)
            from .. import random_jax
            ns = info["noisy_state"]
            t0 = int(ns.time) if hasattr(ns, "pos") else int(dstate.packed[25:26].view(core.torch.int32).item())
            f0 = np.asarray(ns.f_disturb) if hasattr(ns, "pos") else dstate.packed[13:16].cpu().numpy()
            tab = self.env.rollout_disturbance_table(key, env_params, t0, f0, key_mode, True, rng=random_jax, H=self.H)
            return core.torch.from_numpy(tab[None]).to(core.device)

        if self.mode == "online":  # optimal Sigma (covo.py:205-208); get_hessian receives the RAW rng_act (covo.py:205)
            tab_h = table(rng_act, _lib.DISTURB_KEYS_HESSIAN)
            R = core.hessian(dstate.packed, dstate, params_c, a_mean, f_steps=tab_h)
            Sigma, L = core.sigma(R, control_params.sample_sigma)
            a_cov, L = Sigma[0], L[0]
        else:
            t_idx = dstate.packed[25:26].view(core.torch.int32).long()  # env_state.time, stays on the device
            t_idx = t_idx.clamp(0, control_params.a_cov_offline.shape[0] - 1)  # JAX gather clamps
            a_cov = control_params.a_cov_offline.index_select(0, t_idx)[0]
            L = control_params.a_chol_offline.index_select(0, t_idx)[0]
        control_params = control_params.replace(a_cov=a_cov)
        if self.noise_stream == "jax":  # covo.py:212-220 on jax's own threefry stream
            from .. import random_jax
            rng_act, act_key = random_jax.split(rng_act)
            core.randn_jax(act_key)
        else:
            rng_act, act_key = crandom.split(rng_act)  # covo.py:212-224
            core.randn(act_key)
        core.noise_gemm(L, a_mean)
        # covo.py:225-263: deterministic=True -> the gaussian model is off; periodic / mixed still draw from step_key
        rng_act, step_key = random_jax.split(rng_act) if jax_stream else crandom.split(rng_act)
        tab_r = table(step_key, _lib.DISTURB_KEYS_SHARED)
        core.rollout(dstate, params_c, (0.0, 0.0, 0.0), core.compute_info, f_steps=tab_r)
        a_mean_new = core.update(a_mean, control_params.gamma_mean).view(self.H, 4)  # covo.py:266-278
        control_params = control_params.replace(a_mean=a_mean_new)
        out_info = core.info(dstate) if core.compute_info else {}
        return a_mean_new[0], control_params, out_info
