"""MPPI controller behind quadjax's call signature: quadjax/controllers/mppi.py:11-134.

Same constructor / __call__ arguments and return triple; the body is the HIP pipeline
(noise -> fused rollout -> two-stage softmax reduction) instead of jit-compiled JAX.
"""
from __future__ import annotations

import numpy as np

import dataclasses
from dataclasses import dataclass
from typing import Any

from ..dynamics.dataclass import as_device_state
from .base import BaseController
from ._core import SamplingCore


@dataclass(frozen=True)
class MPPIParams:
    """mppi.py:11-19.  a_mean (H,4), a_cov (H,4,4): torch fp32 tensors on the GPU."""
    gamma_mean: float
    gamma_sigma: float
    discount: float
    sample_sigma: float
    a_mean: Any
    a_cov: Any

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


class MPPIController(BaseController):
    def __init__(self, env, control_params, N: int, H: int, lam: float, *, device=None, process_group=None,
                 compute_info: bool = True, propagate_nan=None) -> None:
        super().__init__(env, control_params)
        self.N, self.H, self.lam = N, H, lam
        self.materialize_eps = False  # True: epsilon is written to HBM and the kernels are called one by one (parity)
        self.noise_stream = "philox"  # "jax": sampling keys / epsilon from jax.random's own bitstream (random_jax.py)
        self.alias_outputs = False    # True: returned a_mean / a_cov alias the controller's buffers (no clones)
        self._params_c(env.default_params)  # raises now if env.reward_fn / disturb_type is not one the kernels evaluate
        # mppi.py:119-125's covariance adaptation (gamma_sigma != 0; quadjax's own factory fixes 0, envs/quadrotor.py:715): on
        # sample-sharded ranks the rank record then also carries the weighted second moments (still ONE exchange per step)
        self.core = SamplingCore(N, H, lam, control_params.discount, device=device, process_group=process_group,
                                 compute_info=compute_info, trust_clipped=True,
                                 cov_records=float(getattr(control_params, "gamma_sigma", 0.0)) != 0.0, propagate_nan=propagate_nan)

    def _check_gamma_sigma(self, control_params):
        """A controller built with gamma_sigma = 0 exchanges the 516-float records: on sharded ranks a later gamma_sigma != 0 needs
        a controller built for it (the record size is fixed at construction: exchange buffers, captured graphs)."""
        if control_params.gamma_sigma != 0.0 and self.core.world > 1:
            self.core._need_cov_records()

    def run_episode(self, episode, env_params, control_params, rng, n_steps):
        """n_steps closed-loop steps (this controller + the device env step) enqueued by one C call; keys threaded like
        eval_env's run_one_step.  -> (control_params with the final mean / shifted covariances, rng)."""
        from .. import _lib
        self._check_gamma_sigma(control_params)
        am, cov, rng = self.core.run_episode(_lib.MODE_MPPI, episode, self._params_c(env_params), control_params.a_mean, rng,
                                             n_steps, a_cov=control_params.a_cov, gamma_mean=control_params.gamma_mean,
                                             sample_sigma=control_params.sample_sigma, rollout_deterministic=False,
                                             gamma_sigma=control_params.gamma_sigma)
        a_mean = am.view(self.H, 4)
        if not self.alias_outputs:
            a_mean, cov = a_mean.clone(), cov.clone()
        return control_params.replace(a_mean=a_mean, a_cov=cov), rng

    def __call__(self, obs, env_state, env_params, rng_act, control_params: MPPIParams, info):
        from .. import random as crandom
        core = self.core
        torch = core.torch
        self._check_gamma_sigma(control_params)
        dstate = as_device_state(info["noisy_state"], core.device)  # mppi.py:40
        if self.noise_stream not in ("philox", "jax"):
            raise ValueError(f"noise_stream={self.noise_stream!r}")
        if not self.materialize_eps and self.noise_stream == "philox":
            # ---- production path: one C call / one hipGraph replay (csrc/step.hip)
            from .. import _lib
            # rng_act, act_key = split(rng_act) (mppi.py:53); rng_act, step_key = split(rng_act) and what every sample/step
            # draws from the ONE shared step_key (mppi.py:69,74, deterministic=False: the gaussian vector -- env.rollout_disturbance
            # is its host restatement --, the periodic / mixed redraw) are derived on the device (step.hip: step_begin_kernel,
            # disturb.hip)
            am, cov = core.step(_lib.MODE_MPPI, dstate, self._params_c(env_params), control_params.a_mean, rng_act,
                                a_cov=control_params.a_cov, gamma_mean=control_params.gamma_mean,
                                sample_sigma=control_params.sample_sigma, want_stats=core.compute_info,
                                derive_keys=True, rollout_deterministic=False, gamma_sigma=control_params.gamma_sigma)
            a_mean_new = am.view(self.H, 4)
            if not self.alias_outputs:
                a_mean_new, cov = a_mean_new.clone(), cov.clone()
            control_params = control_params.replace(a_mean=a_mean_new, a_cov=cov)
            out_info = core.info(dstate) if core.compute_info else {}
            return a_mean_new[0], control_params, out_info

        # ---- kernel-by-kernel path with epsilon materialised in HBM (identical values; parity/debug)
        a_mean = core.shift_mean(control_params.a_mean.reshape(-1)).view(self.H, 4)  # mppi.py:43-49
        a_cov = torch.cat([control_params.a_cov[1:], control_params.a_cov[-1:]], dim=0).contiguous()
        control_params = control_params.replace(a_mean=a_mean, a_cov=a_cov)
        Ls = core.cholesky(a_cov, 4, self.H)
        if self.noise_stream == "jax":  # mppi.py:53-60 on jax's own threefry stream
            from .. import random_jax
            rng_act, act_key = random_jax.split(rng_act)
            core.randn_jax(act_key, mppi=True)
        else:
            rng_act, act_key = crandom.split(rng_act)  # mppi.py:53-66
            core.randn(act_key)
        core.noise_blockdiag(Ls, a_mean)
        from .. import _lib
        params_c = self._params_c(env_params)
        if self.noise_stream == "jax":  # the step key and the shared gaussian draw from jax's bitstream too (mppi.py:69,74)
            rng_act, step_key = random_jax.split(rng_act)
            if params_c.disturb_kind in _lib.TABLE_DISTURB_KINDS:
                # the device table kernel draws from the Philox stream: under jax's stream the same table is built on the host
                # with jax.random's key splits and uniforms (envs/quadrotor.py: rollout_disturbance_table) and uploaded
                ns = info["noisy_state"]
                t0 = int(ns.time) if hasattr(ns, "pos") else int(dstate.packed[25:26].view(torch.int32).item())
                f0 = np.asarray(ns.f_disturb) if hasattr(ns, "pos") else dstate.packed[13:16].cpu().numpy()
                f_shared = (0.0, 0.0, 0.0)
                tab = torch.from_numpy(self.env.rollout_disturbance_table(step_key, env_params, t0, f0, _lib.DISTURB_KEYS_SHARED,
                                                                          False, rng=random_jax, H=self.H)[None]).to(core.device)
            else:
                f_shared, tab = self.env.rollout_disturbance(step_key, env_params, deterministic=False, rng=random_jax), None
        else:
            rng_act, step_key = crandom.split(rng_act)  # mppi.py:69-106
            if params_c.disturb_kind in _lib.TABLE_DISTURB_KINDS:  # periodic / sin / drag / mixed (free.py:10-58)
                f_shared = (0.0, 0.0, 0.0)
                tab = core.disturb_table(params_c, dstate.packed, key=step_key, key_mode=_lib.DISTURB_KEYS_SHARED,
                                         deterministic=False)
            else:
                f_shared, tab = self.env.rollout_disturbance(step_key, env_params, deterministic=False), None
        core.rollout(dstate, params_c, f_shared, core.compute_info, f_steps=tab)
        if control_params.gamma_sigma != 0.0:  # mppi.py:109-125: mean, then the covariance about the NEW mean
            a_mean_new, a_cov_new = core.update_cov(a_mean.reshape(-1), control_params.gamma_mean, a_cov, control_params.gamma_sigma)
            control_params = control_params.replace(a_mean=a_mean_new.view(self.H, 4), a_cov=a_cov_new)
            a_mean_new = a_mean_new.view(self.H, 4)
        else:
            a_mean_new = core.update(a_mean.reshape(-1), control_params.gamma_mean).view(self.H, 4)  # mppi.py:109-118
            control_params = control_params.replace(a_mean=a_mean_new)
        out_info = core.info(dstate) if core.compute_info else {}
        return a_mean_new[0], control_params, out_info
