"""CoVO-online for E independent env instances in one call (BASELINE.json configs[4]).

The reference runs its `--mode render` / eval loop on one env instance and reaches many instances through
`jax.vmap` of the whole controller (quadjax/envs/quadrotor.py:497-538 is written per instance and vmap-clean).
Here that is `covo_mpc_step_batched` (include/covo_hip.h, csrc/step.hip): one hipGraph holding ONE batched
Hessian + Sigma launch set for all instances and the per-instance sampling path.  Every instance has its own
state, reference trajectory, (domain-randomised) parameters, mean and key; instance e's result is bit-identical
to `CoVOController.__call__` on that instance alone (tests/test_gpu_parity.py::test_batched_step_equals_replicas).
Instances never exchange data ("replicas only"): to use G GPUs give each rank E/G instances.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib
from .._lib import COVO_H, COVO_NA, check
from ..dynamics.dataclass import as_device_state
from ._core import SamplingCore


class BatchedCoVOController:
    def __init__(self, env, n_envs: int, N: int, H: int, lam: float, *, discount: float = 1.0, gamma_mean: float = 1.0,
                 sample_sigma: float = 0.5, a_mean_init=None, device=None):
        if not 0 < n_envs <= _lib.COVO_MAX_ENVS:
            raise ValueError(f"n_envs={n_envs} outside (0, {_lib.COVO_MAX_ENVS}]")
        self.env, self.E, self.N, self.H = env, int(n_envs), int(N), int(H)
        self.gamma_mean, self.sample_sigma = float(gamma_mean), float(sample_sigma)
        self.rollover_terminate = not getattr(env, "disable_rollover_terminate", True)  # quadrotor.py:486
        # every disturbance model of the env is taken: for periodic / sin / drag / mixed covo_mpc_step_batched builds each
        # instance's per-step tables (csrc/disturb.hip) from its state, raw key and disturb_params inside the graph
        # one call advances all instances: the ~56 launches are worth a graph (same GPU time as eager, 40 us instead of
        # 150-270 us of host time per call)
        self.core = SamplingCore(N, H, lam, discount, device=device, compute_info=False, trust_clipped=True, use_graph=True)
        torch = self.core.torch
        f32 = dict(dtype=torch.float32, device=self.core.device)
        E, n = self.E, self.N
        self.a_mean = torch.zeros((E, COVO_NA), **f32)
        if a_mean_init is not None:
            self.a_mean.copy_(torch.as_tensor(a_mean_init, **f32).reshape(1, COVO_NA).expand(E, COVO_NA))
        self.a_cov = torch.zeros((E, COVO_NA, COVO_NA), **f32)
        self._a = torch.empty((E, COVO_H, n, 4), **f32)
        self._cost = torch.empty((E, n), **f32)
        self._groupmin = torch.empty((E, (n + 63) // 64), **f32)
        self._states = torch.zeros((E, _lib.COVO_STATE_FLOATS), **f32)
        self._traj = None  # (pos [E,T,3], vel [E,T,3], source ptrs)
        self._params = None
        self._args = None

    def set_instances(self, env_states, env_params):
        """Bind the E instances' reference trajectories and parameters (once per episode)."""
        torch = self.core.torch
        ds = [as_device_state(s, self.core.device) for s in env_states]
        T = ds[0].T
        if any(d.T != T for d in ds):
            raise ValueError("all instances must share the trajectory length T")
        pos = torch.stack([d.pos_traj.reshape(T, 3) for d in ds]).contiguous()
        vel = torch.stack([d.vel_traj.reshape(T, 3) for d in ds]).contiguous()
        self._traj = (pos, vel, T)
        from .base import env_model_params_c
        self._params = (_lib.EnvParamsC * self.E)(*[env_model_params_c(self.env, p) for p in env_params])
        a = _lib.BatchArgsC()
        a.n_envs, a.n_samples, a.T = self.E, self.N, T
        a.states, a.pos_traj, a.vel_traj = self._states.data_ptr(), pos.data_ptr(), vel.data_ptr()
        a.a_mean, a.a_cov = self.a_mean.data_ptr(), self.a_cov.data_ptr()
        a.a, a.cost, a.groupmin = self._a.data_ptr(), self._cost.data_ptr(), self._groupmin.data_ptr()
        a.gamma_mean, a.sample_sigma = self.gamma_mean, self.sample_sigma
        self._args = a
        self._states_buf = self._states
        self._episode = None

    def __call__(self, noisy_states, rng_acts):
        """One control step of every instance.  noisy_states: E env states (or a float32 [E, 32] device tensor of
        packed states); rng_acts: uint32 [E, 2] raw controller keys.  -> first actions [E, 4] (view of a_mean)."""
        if self._args is None:
            raise RuntimeError("call set_instances(env_states, env_params) first")
        torch = self.core.torch
        buf = self._states_buf  # the tensor args.states points at: the controller's own, or a bound episode's noisy states
        if noisy_states is None or noisy_states is buf:
            pass  # a bound BatchedDeviceEpisode: the env step kernel has already written them
        elif torch.is_tensor(noisy_states):
            buf.copy_(noisy_states, non_blocking=True)
        else:
            packed = [as_device_state(s, self.core.device).packed for s in noisy_states]
            torch.stack(packed, out=buf)
        keys = np.ascontiguousarray(np.asarray(rng_acts, dtype=np.uint32).reshape(self.E, 2))
        check(self.core.lib.covo_mpc_step_batched(self.core.h, C.byref(self._args), self._params,
                                                  keys.ctypes.data_as(C.POINTER(C.c_uint32)), self.core.stream()),
              "covo_mpc_step_batched")
        return self.a_mean.view(self.E, COVO_H, 4)[:, 0]

    def time_phases(self, step_mask: int, reps: int = 10) -> float:
        """GPU microseconds of the selected launch groups of the LAST batched step (2 Hessian, 4 Sigma, 8 GEMM, 16 rollout,
        32 update), replayed `reps` times from one graph (covo_debug_time_batched)."""
        us = C.c_float(0.0)
        self.core.torch.cuda.synchronize()
        check(self.core.lib.covo_debug_time_batched(self.core.h, int(step_mask), int(reps), C.byref(us), self.core.stream()),
              "covo_debug_time_batched")
        return float(us.value)

    def bind_episode(self, episode):
        """Plan from a BatchedDeviceEpisode's buffers: its noisy states ARE the controller's input (rewritten by every env step on
        the device), its trajectories and parameters the instances' (once per episode)."""
        if episode.E != self.E:
            raise ValueError(f"episode has {episode.E} instances, controller {self.E}")
        self._traj = (episode.pos_traj, episode.vel_traj, episode.T)
        self._params = episode.params_c
        a = _lib.BatchArgsC()
        a.n_envs, a.n_samples, a.T = self.E, self.N, episode.T
        a.states, a.pos_traj, a.vel_traj = episode.noisy.data_ptr(), episode.pos_traj.data_ptr(), episode.vel_traj.data_ptr()
        a.a_mean, a.a_cov = self.a_mean.data_ptr(), self.a_cov.data_ptr()
        a.a, a.cost, a.groupmin = self._a.data_ptr(), self._cost.data_ptr(), self._groupmin.data_ptr()
        a.gamma_mean, a.sample_sigma = self.gamma_mean, self.sample_sigma
        self._args = a
        self._states_buf = episode.noisy
        self._episode = episode

    def run_episode(self, episode, rngs, n_steps: int):
        """covo_run_episode_batched: n_steps x { batched control step on the noisy states -> batched env step }, all enqueued by
        ONE C call; every instance's key chain threaded like eval_env's run_one_step (quadrotor.py:520-538).  rngs: uint32
        [E, 2] -> the chains' keys after the segment.  Asynchronous; episode.read_log() synchronises."""
        if getattr(self, "_episode", None) is not episode:
            self.bind_episode(episode)
        env = self.env
        keys = np.ascontiguousarray(np.asarray(rngs, dtype=np.uint32).reshape(self.E, 2)).copy()
        check(self.core.lib.covo_run_episode_batched(
            self.core.h, C.byref(self._args), self._params, _lib.ptr(episode.true), _lib.ptr(episode.acc_traj),
            1 if env.generate_noisy_state else 0, float(env.default_params.obs_noise_scale), _lib.ptr(episode.log),
            int(episode.log.shape[1]), int(episode.n_steps), keys.ctypes.data_as(C.POINTER(C.c_uint32)), int(n_steps),
            self.core.stream()), "covo_run_episode_batched")
        episode.n_steps += int(n_steps)
        return keys
