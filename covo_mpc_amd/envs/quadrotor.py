"""Quadrotor environment plumbing + experiment drivers: quadjax/envs/quadrotor.py.

Host-side (numpy fp32) mirror of `Quad3D` (quadrotor.py:23-503) -- same constructor, attributes,
`step_env` / `raw_step` / `reset_env` / `get_info` / `get_obs*` / `is_terminal` -- plus
`get_controller` (:670-752), `eval_env` (:506-591), `Args` (:755-766) and `main` (:769-803).
No kernels live here: one state per control step is serial scalar math; the
N x H rollouts the controllers need run in csrc/rollout.hip.  The controllers receive
`info["noisy_state"]` exactly like the reference's (covo.py:198).
"""
from __future__ import annotations

import argparse
import os
import pickle
import time as time_module
from dataclasses import dataclass as pydataclass
from functools import partial

import numpy as np

from .. import controllers
from .. import random as crandom
from ..dynamics import utils
from ..dynamics.dataclass import Action3D, DeviceState, EnvParams3D, EnvState3D
from ..dynamics.free import get_quadrotor_1st_order_dyn
from .base import BaseEnvironment

f32 = np.float32


class Quad3D(BaseEnvironment):
    def __init__(self, task: str = "tracking", obs_type: str = "quad", enable_randomizer: bool = True,
                 lower_controller: str = "base", disturb_type: str = "periodic",
                 disable_rollover_terminate: bool = False, generate_noisy_state: bool = False, device=None):
        super().__init__()
        self.task = task
        self.disable_rollover_terminate = disable_rollover_terminate
        self.generate_noisy_state = generate_noisy_state
        self.disturb_type = disturb_type
        self.device = device
        dp = self.default_params
        gens = {  # quadrotor.py:49-84
            "tracking": (utils.generate_lissa_traj, utils.tracking_penyaw_reward_fn),
            "tracking_slow": (utils.generate_lissa_traj_slow, utils.tracking_realworld_reward_fn),
            "tracking_zigzag": (utils.generate_zigzag_traj, utils.tracking_penyaw_reward_fn),
            "hovering": (utils.generate_fixed_traj, utils.tracking_penyaw_reward_fn),
        }
        if task not in gens:
            raise NotImplementedError(task)
        gen, self.reward_fn = gens[task]
        self.generate_traj = partial(gen, dp.max_steps_in_episode, dp.dt)
        self.get_init_state = self.get_zero_state
        self.step_fn, self.dynamics_fn = get_quadrotor_1st_order_dyn(disturb_type=disturb_type)  # :87-89
        self.get_err_pos = lambda state: np.linalg.norm(state.pos_tar - state.pos)
        self.get_err_vel = lambda state: np.linalg.norm(state.vel_tar - state.vel)
        if lower_controller != "base":  # the L1 lower controllers live on the reference's `rl` branch only
            raise NotImplementedError(lower_controller)
        self.default_control_params = 0.0
        self.control_fn = lambda obs, state, env_params, rng_act, input_action: (input_action, None, state)
        self.sim_dt = dp.dt
        self.substeps = 1
        self.enable_randomizer = enable_randomizer
        if enable_randomizer and "params" not in obs_type:
            print("Warning: enable domain randomziation without params in obs_type")
        if obs_type == "quad_params":
            self.get_obs = self.get_obs_quad_params
            self.obs_dim = 39 + dp.traj_obs_len * 6
        elif obs_type == "quad":
            self.get_obs = self.get_obs_quadonly
            self.obs_dim = 19 + dp.traj_obs_len * 6
        else:
            raise NotImplementedError(obs_type)
        self.equib = np.array([0.0] * 6 + [1.0] + [0.0] * 9, dtype=f32)
        self.action_dim = 4
        self.adapt_obs_dim = 22 * dp.adapt_horizon
        self.param_obs_dim = 20

    @property
    def default_params(self) -> EnvParams3D:
        return EnvParams3D()

    # ---- domain randomisation (quadrotor.py:132-171) ------------------------------------------------
    def sample_params(self, key) -> EnvParams3D:
        p = self.default_params
        if self.enable_randomizer:
            param_key = crandom.split(key)[0]
            r = crandom.uniform(param_key, (17,), -1.0, 1.0)
            I_diag = p.I_diag_mean + r[1:4] * p.I_diag_std
            return EnvParams3D(m=float(p.m_mean + r[0] * p.m_std), I=np.diag(I_diag).astype(f32),
                               action_scale=float(p.action_scale_mean + r[4] * p.action_scale_std),
                               alpha_bodyrate=float(p.alpha_bodyrate_mean + r[5] * p.alpha_bodyrate_std),
                               disturb_params=(r[6:12] * p.disturb_scale).astype(f32))
        return EnvParams3D(disturb_params=crandom.uniform(key, (6,), -1.0, 1.0))

    # ---- key methods ----------------------------------------------------------------------------------
    def step_env(self, key, state: EnvState3D, action, params: EnvParams3D, deterministic: bool = False,
                 need_info: bool = True):
        """quadrotor.py:215-248: reward/done of the PRE-step state, one raw_step."""
        action = np.clip(np.asarray(action, dtype=f32), -1.0, 1.0)
        params = params.replace(dyn_noise_scale=params.dyn_noise_scale * (1.0 - float(deterministic)))  # :234-235
        sub_action, _, state = self.control_fn(None, state, params, key, action)
        next_state = self.raw_step(key, state, sub_action, params)
        reward = self.reward_fn(state, params)
        done = bool(self.is_terminal(state, params))
        if not need_info:
            return None, next_state, reward, done, None
        info_key, key = crandom.split(key)
        info = self.get_info(info_key, state, next_state, params)
        return self.get_obs(next_state, params), next_state, reward, done, info

    def raw_step(self, key, state, sub_action, params):
        """quadrotor.py:250-263."""
        sub_action = np.clip(np.asarray(sub_action, dtype=f32), -1.0, 1.0)
        thrust = (sub_action[0] + f32(1.0)) / f32(2.0) * f32(params.max_thrust)
        torque = sub_action[1:] * params.max_torque
        key, step_key = crandom.split(key)
        return self.step_fn(params, state, Action3D(thrust=thrust, torque=torque.astype(f32)), step_key, self.sim_dt)

    def rollout_disturbance(self, step_key, params, deterministic: bool, rng=crandom):
        """The single f_disturb vector every sample/step of a controller rollout receives under 'none' / 'gaussian': all
        N x H step_env calls share `step_key` (covo.py:225,231 / mppi.py:69,74), so free.py:147's draw is one vector.  Key
        derivation follows step_env -> raw_step -> step_fn (quadrotor.py:262, free.py:136,144); `rng`: the key module
        (covo_mpc_amd.random, or random_jax for jax's own bitstream).  The other models (periodic / sin / drag / mixed) are
        per-step or per-sample: csrc/disturb.hip builds their table on the device (SamplingCore.disturb_table)."""
        if self.disturb_type == "none":
            return np.zeros(3, dtype=f32)
        if self.disturb_type == "gaussian":
            scale = params.dyn_noise_scale * (1.0 - float(deterministic))
            _, k = rng.split(step_key)      # raw_step: key, step_key = split(key)
            k, _ = rng.split(k)             # step_fn:  key, key_dyn = split(key)
            disturb_key, _ = rng.split(k)   #           disturb_key, key = split(key)
            return (f32(scale) * np.asarray(rng.normal(disturb_key, (3,)), dtype=f32)).astype(f32)
        raise NotImplementedError(f"disturb_type={self.disturb_type!r}: no single shared vector (use SamplingCore.disturb_table)")

    def rollout_disturbance_table(self, key, params, time: int, f_disturb0, key_mode: int, deterministic: bool, rng=crandom,
                                  H: int = 32):
        """Host restatement of csrc/disturb.hip: the per-step disturbance table of a controller rollout, float32 [H][4] with
        row k = {g_k[3], c_k}, f_k = c_drag drag(vel_{k-1}) + c_k f_{k-1} + g_k for k >= 1 (row 0 unused: step 0 integrates the
        state's own f_disturb).  `rng`: the key module -- covo_mpc_amd.random (then the rows are the device kernel's), or
        random_jax for jax's own bitstream (noise_stream = "jax": keys split and uniforms / normals drawn as jax.random does).
        key_mode (covo_mpc_amd._lib.DISTURB_KEYS_*): who calls step_env with which key -- SHARED: every step the same `key`
        (covo.py:225,231; mppi.py:69,74); HESSIAN: rng_k, key = split(key) per step (covo.py:150-153); NOMINAL: _, key = split(key);
        rng_step, key = split(key) (covo.py:58-70)."""
        from .. import _lib
        kind = self.disturb_type
        tab = np.zeros((H, 4), dtype=f32)
        dp = np.asarray(params.disturb_params, dtype=f32)
        period = int(params.disturb_period) if int(params.disturb_period) > 0 else 1
        scale = f32(params.disturb_scale)
        held = np.asarray(f_disturb0, dtype=f32).copy()
        key = np.asarray(key)

        def disturb_key(k):  # raw_step: key, step_key = split(k); step_fn: key, key_dyn = split(step_key); disturb_key, key = split(key)
            _, a = rng.split(k)
            b, _ = rng.split(a)
            d, _ = rng.split(b)
            return d

        def sin_term(t):  # free.py:27-38 in fp32, like the env
            amp = dp[:3] * scale
            per = dp[:3] * f32(period / 3) + f32(period)
            return (amp * np.sin(f32(2 * np.pi) / per * f32(t) + dp[3:6] * f32(2 * np.pi))).astype(f32)

        for k in range(H - 1):
            if key_mode == _lib.DISTURB_KEYS_HESSIAN:
                sk, key = rng.split(key)
            elif key_mode == _lib.DISTURB_KEYS_NOMINAL:
                _, key = rng.split(key)
                sk, key = rng.split(key)
            else:
                sk = key
            t = int(time) + k
            hit = t % period == 0
            g, c = np.zeros(3, dtype=f32), f32(0.0)
            if kind == "gaussian":
                if not deterministic:
                    g = (f32(params.dyn_noise_scale) * np.asarray(rng.normal(disturb_key(sk), (3,)), dtype=f32)).astype(f32)
            elif kind == "periodic":
                if hit:
                    held = np.asarray(rng.uniform(disturb_key(sk), (3,), -scale, scale), dtype=f32)
                g = held.copy()
            elif kind == "sin":
                g = sin_term(t)
            elif kind == "mixed":
                u = np.asarray(rng.uniform(disturb_key(sk), (3,), -scale, scale), dtype=f32) if hit else np.zeros(3, dtype=f32)
                g = ((sin_term(t) + u) / f32(3)).astype(f32)
                c = f32(0.0) if hit else f32(1.0) / f32(3.0)
            tab[k + 1, :3] = g
            tab[k + 1, 3] = c
        return tab

    def get_zero_state(self, key, params) -> EnvState3D:
        """quadrotor.py:265-312."""
        traj_key, disturb_key, key = crandom.split(key, 3)
        pos_traj, vel_traj, acc_traj = self.generate_traj(traj_key)
        z3 = np.zeros(3, dtype=f32)
        hist = self.default_params.adapt_horizon + 2
        traj_dev = None
        if self.device is not None:
            import torch
            traj_dev = (torch.from_numpy(pos_traj).to(self.device), torch.from_numpy(vel_traj).to(self.device))
        return EnvState3D(
            pos=z3.copy(), vel=z3.copy(), omega=z3.copy(), omega_tar=z3.copy(), quat=np.array([0, 0, 0, 1], dtype=f32),
            pos_tar=pos_traj[0], vel_tar=vel_traj[0], acc_tar=acc_traj[0],
            pos_traj=pos_traj, vel_traj=vel_traj, acc_traj=acc_traj, last_thrust=0.0, last_torque=z3.copy(), time=0,
            f_disturb=crandom.uniform(disturb_key, (3,), -params.disturb_scale, params.disturb_scale),
            vel_hist=np.zeros((hist, 3), f32), omega_hist=np.zeros((hist, 3), f32), action_hist=np.zeros((hist, 4), f32),
            control_params=self.default_control_params, traj_dev=traj_dev)

    def get_info(self, rng, state, next_state, params) -> dict:
        """quadrotor.py:314-361."""
        if self.generate_noisy_state:
            rng_pos, rng_vel, rng_quat, rng_omega, rng = crandom.split(rng, 5)
            s = f32(self.default_params.obs_noise_scale)
            noisy_state = next_state.replace(
                pos=(next_state.pos + crandom.normal(rng_pos, (3,)) * s * f32(0.25)).astype(f32),
                vel=(next_state.vel + crandom.normal(rng_vel, (3,)) * s * f32(0.5)).astype(f32),
                quat=(next_state.quat + crandom.normal(rng_quat, (4,)) * s * f32(0.02)).astype(f32),
                omega=(next_state.omega + crandom.normal(rng_omega, (3,)) * s * f32(0.5)).astype(f32))
        else:
            noisy_state = None
        return {"discount": 1.0, "err_pos": self.get_err_pos(state), "err_vel": self.get_err_vel(state),
                "obs_param": self.get_obs_paramsonly(state, params), "noisy_state": noisy_state}

    def reset_env(self, key, params):
        """quadrotor.py:363-370 (note the (obs, info, state) order)."""
        state = self.get_init_state(key, params)
        info_key, key = crandom.split(key)
        info = self.get_info(info_key, state, state, params)
        return self.get_obs(state, params), info, state

    def get_obs_quadonly(self, state, params):
        """quadrotor.py:372-394."""
        dp = self.default_params
        idx = np.clip(state.time + 1 + np.arange(dp.traj_obs_len) * dp.traj_obs_gap, 0, state.pos_traj.shape[0] - 1)
        return np.concatenate([state.pos, state.vel / 3.0, state.quat, state.omega / 5.0, state.pos_tar,
                               state.vel_tar / 3.0, state.pos_traj[idx].flatten(),
                               state.vel_traj[idx].flatten() / 3.0]).astype(f32)

    def get_obs_paramsonly(self, state, params):
        """quadrotor.py:425-452."""
        return np.concatenate([
            (np.diag(params.I) - params.I_diag_mean) / params.I_diag_std, state.f_disturb / params.disturb_scale,
            (params.hook_offset - params.hook_offset_mean) / params.hook_offset_std, params.disturb_params,
            [(params.m - params.m_mean) / params.m_std,
             (params.action_scale - params.action_scale_mean) / params.action_scale_std,
             (params.alpha_bodyrate - params.alpha_bodyrate_mean) / params.alpha_bodyrate_std]]).astype(f32)

    def get_obs_quad_params(self, state, params):
        """quadrotor.py:465-470."""
        return np.concatenate([self.get_obs_quadonly(state, params), self.get_obs_paramsonly(state, params)])

    def is_terminal(self, state, params) -> bool:
        """quadrotor.py:479-490."""
        done = (state.time >= params.max_steps_in_episode) or bool(np.any(np.abs(state.pos) > 3.0))
        if not self.disable_rollover_terminate:
            done = done or (state.quat[3] < np.cos(np.pi / 4.0)) or bool(np.any(np.abs(state.omega) > 100.0))
        return done


class DeviceEpisode:
    """One episode whose env state lives on the device: the true state, its noisy copy (what the
    controller plans from), the reference trajectory and the per-step log {reward, err_pos, err_vel, done}.
    `reset` is host plumbing (trajectory generation, quadrotor.py:265-312) followed by one upload; `step` launches
    covo_env_step, which derives the five noise keys of `Quad3D.step` from the step key on the device -- no sync,
    so an episode is 300 x (controller graph + this launch) with one read-back of the log at the end.
    auto_reset (default, as BaseEnvironment.step: quadjax/envs/base.py:22-40): when the pre-step state is terminal the kernel
    stores reset_env(key_reset)'s state, noisy copy and a NEW trajectory (written over pos_traj / vel_traj / acc_traj in place);
    the log row then carries the reset state's errors and done = 1.  `state0` keeps the first reset's host state."""

    def __init__(self, env: "Quad3D", key, params, lib_handle, device, auto_reset: bool = True):
        import torch
        from .. import _lib
        self.env, self.params, self.device = env, params, device
        self.lib, self.h = lib_handle
        self._lib = _lib
        obs, info, state = env.reset(key, params)
        self.state0 = state
        ns = info["noisy_state"] if info["noisy_state"] is not None else state
        self.true = torch.from_numpy(state.pack()).to(device)
        self.noisy = torch.from_numpy(ns.pack()).to(device)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        self.pos_traj, self.vel_traj, self.acc_traj = up(state.pos_traj), up(state.vel_traj), up(state.acc_traj)
        self.T = int(state.pos_traj.shape[0])
        self.log = torch.zeros((params.max_steps_in_episode + 1, 4), dtype=torch.float32, device=device)
        from ..controllers.base import env_model_params_c
        # incl. the env's reward, disturbance model and reset generator (env_step.hip runs all of them)
        self.params_c = env_model_params_c(env, params, auto_reset=auto_reset)
        self.n_steps = 0

    @property
    def noisy_state(self) -> DeviceState:
        return DeviceState(packed=self.noisy, pos_traj=self.pos_traj, vel_traj=self.vel_traj, time=None)

    @staticmethod
    def leaf_keys(step_key):
        """(disturb, pos, vel, quat, omega) keys of Quad3D.step(step_key, ...): base.py:22 -> step_env (split for
        raw_step and for get_info from the same key, quadrotor.py:262 / :246) -> free.py:136,144 -> quadrotor.py:324."""
        k = crandom.split(step_key)[0]            # base.step: key, key_reset = split(key)
        info_key, raw_key = crandom.split(k)      # step_env: info_key = split(key)[0]; raw_step: step_key = split(key)[1]
        dk = crandom.split(crandom.split(raw_key)[0])[0]   # step_fn: key, key_dyn = split(key); disturb_key, key = split(key)
        return np.concatenate([dk[None], crandom.split(info_key, 5)[:4]]).astype(np.uint32)

    def step(self, step_key, action, stream=None):
        """action: float32[4] device tensor (the controller's u).  Asynchronous."""
        import ctypes as C
        import torch
        keys = np.ascontiguousarray(np.asarray(step_key).reshape(-1)[:2], dtype=np.uint32)  # leaf keys derived on the device
        ptr = self._lib.ptr
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream) if stream is None else stream
        self._lib.check(self.lib.covo_env_step(
            self.h, ptr(self.true), ptr(self.noisy), ptr(self.pos_traj), ptr(self.vel_traj), ptr(self.acc_traj), self.T,
            C.byref(self.params_c), ptr(action), keys.ctypes.data_as(C.POINTER(C.c_uint32)),
            1 if self.env.generate_noisy_state else 0, float(self.env.default_params.obs_noise_scale), ptr(self.log),
            self.n_steps, st), "covo_env_step")
        self._keep = (keys, action)
        self.n_steps += 1

    def read_log(self):
        """-> float32[n_steps, 4] = reward, err_pos, err_vel, done (pre-step state); synchronises, then checks the handle's
        sticky device status: a kernel of the enqueued steps that failed (a starved grid barrier of the Sigma chain, see
        SamplingCore.shared_device) has poisoned every later step of the segment -- raise instead of returning its log."""
        out = self.log[:self.n_steps].cpu().numpy()
        st = int(self.lib.covo_device_status(self.h, 0))
        if st != 0:
            raise self._lib.CovoError(f"device status 0x{st:x} after the episode segment (covo_device_status): a kernel of an "
                                      "enqueued step failed; its mean / Sigma and everything after it are invalid.  "
                                      "Build the controller with shared_device=True when the GPU is shared.")
        return out


class BatchedDeviceEpisode:
    """E independent env instances on the device (BASELINE configs[4]): instance e has its own (domain-randomised,
    quadrotor.py:132-171) parameters, reset key, true state, noisy copy, reference trajectory and log.  `step` launches
    covo_env_step_batched -- every instance's Quad3D.step in ONE launch; BatchedCoVOController.run_episode enqueues whole
    episodes (control step + env step for all instances) from one C call.  auto_reset: as DeviceEpisode, per instance."""

    def __init__(self, env: "Quad3D", keys, params_list, lib_handle, device, auto_reset: bool = True):
        import torch
        from .. import _lib
        from ..controllers.base import env_model_params_c
        self.env, self.params, self.device = env, list(params_list), device
        self.lib, self.h = lib_handle
        self._lib = _lib
        self.E = len(self.params)
        if not 0 < self.E <= _lib.COVO_MAX_ENVS or len(keys) != self.E:
            raise ValueError(f"{self.E} instances (1..{_lib.COVO_MAX_ENVS}), {len(keys)} keys")
        states, noisy = [], []
        for k, p in zip(keys, self.params):
            obs, info, st = env.reset(k, p)
            states.append(st)
            noisy.append(info["noisy_state"] if info["noisy_state"] is not None else st)
        self.states0 = states
        T = int(states[0].pos_traj.shape[0])
        if any(int(s.pos_traj.shape[0]) != T for s in states):
            raise ValueError("all instances must share the trajectory length T")
        if any(p.max_steps_in_episode != self.params[0].max_steps_in_episode for p in self.params):
            raise ValueError("all instances must share max_steps_in_episode")
        self.T = T
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        self.true = up(np.stack([s.pack() for s in states]))
        self.noisy = up(np.stack([s.pack() for s in noisy]))
        self.pos_traj = up(np.stack([s.pos_traj for s in states]))
        self.vel_traj = up(np.stack([s.vel_traj for s in states]))
        self.acc_traj = up(np.stack([s.acc_traj for s in states]))
        self.log = torch.zeros((self.E, self.params[0].max_steps_in_episode + 1, 4), dtype=torch.float32, device=device)
        self.params_c = (_lib.EnvParamsC * self.E)(*[env_model_params_c(env, p, auto_reset=auto_reset) for p in self.params])
        self.n_steps = 0

    def step(self, step_keys, a_mean, stream=None):
        """step_keys: uint32 [E, 2] (the key Quad3D.step receives, per instance); a_mean: float32 [E, 128] device tensor whose first
        four entries per instance are the action.  Asynchronous."""
        import ctypes as C
        import torch
        keys = np.ascontiguousarray(np.asarray(step_keys, dtype=np.uint32).reshape(self.E, 2))
        ptr = self._lib.ptr
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream) if stream is None else stream
        self._lib.check(self.lib.covo_env_step_batched(
            self.h, self.E, ptr(self.true), ptr(self.noisy), ptr(self.pos_traj), ptr(self.vel_traj), ptr(self.acc_traj), self.T,
            self.params_c, ptr(a_mean), keys.ctypes.data_as(C.POINTER(C.c_uint32)), 1 if self.env.generate_noisy_state else 0,
            float(self.env.default_params.obs_noise_scale), ptr(self.log), int(self.log.shape[1]), self.n_steps, st),
            "covo_env_step_batched")
        self._keep = (keys, a_mean)
        self.n_steps += 1

    def read_log(self):
        """-> float32 [E, n_steps, 4] = reward, err_pos, err_vel, done (pre-step states); synchronises and checks the device status."""
        out = self.log[:, :self.n_steps].cpu().numpy()
        st = int(self.lib.covo_device_status(self.h, 0))
        if st != 0:
            raise self._lib.CovoError(f"device status 0x{st:x} after the batched episode segment (covo_device_status)")
        return out


def eval_env_batched(env: Quad3D, n_envs: int, controller_params: str = "N4096_H32_lam0.01", n_steps=None, seed: int = 1, device=None,
                     verbose: bool = True):
    """BASELINE configs[4] as a driver: `n_envs` domain-randomised instances of `env` (each with parameters from
    env.sample_params, its own reset key and key chain, quadrotor.py:132-171 + 506-591 per instance) run one episode under
    covo-online, controller and env on the device, ONE host sync.  -> mean position error per instance [n_envs]."""
    from .. import controllers
    rng = crandom.PRNGKey(seed)
    ks = crandom.split(rng, 3 * n_envs + 1)
    params = [env.sample_params(ks[e]) for e in range(n_envs)]
    N, H, lam = (lambda p: (int(p[0][1:]), int(p[1][1:]), float(p[2][3:])))(controller_params.split("_"))
    c0, cp0 = get_controller(env, "covo-online", controller_params, device=device, compute_info=False)
    cp0 = c0.init_control_params
    b = controllers.BatchedCoVOController(env, n_envs, N, H, lam, discount=cp0.discount, gamma_mean=cp0.gamma_mean,
                                          sample_sigma=cp0.sample_sigma, a_mean_init=cp0.a_mean, device=c0.core.device)
    del c0
    ep = BatchedDeviceEpisode(env, ks[n_envs:2 * n_envs], params, (b.core.lib, b.core.h), b.core.device)
    T = params[0].max_steps_in_episode if n_steps is None else int(n_steps)
    t0 = time_module.time()
    b.run_episode(ep, ks[2 * n_envs:3 * n_envs], T)
    log = ep.read_log()
    if verbose:
        el = time_module.time() - t0
        print(f"{n_envs} instances x {T} steps in {el:.2f}s = {n_envs * T / el:.0f} env-steps/s; "
              f"err_pos mean over instances: {log[:, :, 1].mean():.3f} (min {log[:, :, 1].mean(axis=1).min():.3f}, "
              f"max {log[:, :, 1].mean(axis=1).max():.3f})")
    return log[:, :, 1].mean(axis=1)


def eval_env_device(env: Quad3D, controller, total_steps=30000, num_trajs=4, seed=1, verbose=True):
    """eval_env (quadrotor.py:506-591) with the env step on the device: same key threading, same protocol (num_trajs
    reset keys x episodes x max_steps_in_episode steps, mean/std over episodes of the mean position error), one host
    sync per EPISODE instead of one per step."""
    rng = crandom.PRNGKey(seed)
    T = env.default_params.max_steps_in_episode
    core = controller.core
    keep_alias = getattr(controller, "alias_outputs", False)
    controller.alias_outputs = True  # u stays a view of the controller's mean buffer: nothing leaves the device

    def run_one_ep(rng_reset, rng):
        env_params = env.default_params  # :543 (default params even under DR)
        ep = DeviceEpisode(env, rng_reset, env_params, (core.lib, core.h), core.device)
        rng_control, rng = crandom.split(rng)
        control_params = controller.reset(ep.state0, env_params, controller.init_control_params, rng_control)
        if hasattr(controller, "run_episode") and core.world == 1:
            # the whole episode is enqueued by ONE C call (covo_run_episode), keys threaded as run_one_step does
            control_params, rng = controller.run_episode(ep, env_params, control_params, rng, T)
        else:
            for _ in range(T):  # run_one_step, :520-538
                rng, rng_act, rng_step, rng_control = crandom.split(rng, 4)
                action, control_params, _ = controller(None, None, env_params, rng_act, control_params,
                                                       {"noisy_state": ep.noisy_state})
                ep.step(rng_step, action)
                rng, rng_control = crandom.split(rng)
        log = ep.read_log()
        # info["err_pos"] of step t is the error of the state BEFORE that step (quadrotor.py:352); the host loop
        # records it after each env.step, i.e. rows 0..T-1 of the log
        return rng, log[:, 1]

    t0 = time_module.time()
    num_eps = int(total_steps // T)
    err_pos_ep = []
    rng, rng_reset_meta = crandom.split(rng)
    for i, rng_reset in enumerate(crandom.split(rng_reset_meta, num_trajs)):
        for _ in range(max(num_eps // num_trajs, 1)):
            rng, err_pos = run_one_ep(rng_reset, rng)
            err_pos_ep.append(err_pos.mean())
    controller.alias_outputs = keep_alias
    err_pos_ep = np.asarray(err_pos_ep)
    if verbose:
        print(f"env running time: {time_module.time()-t0:.2f}s")
        print(f"err_pos mean: {err_pos_ep.mean():.3f}, std: {err_pos_ep.std():.3f}")
    return err_pos_ep


def eval_env(env: Quad3D, controller, total_steps=30000, filename="", num_trajs=4, seed=1, save=True, verbose=True):
    """quadrotor.py:506-591: `num_trajs` reset keys x (episodes) x max_steps_in_episode steps; reports the
    mean/std over episodes of the mean position error."""
    rng = crandom.PRNGKey(seed)
    T = env.default_params.max_steps_in_episode

    def run_one_ep(rng_reset, rng):
        env_params = env.default_params  # :543 (default params even under DR)
        obs, info, env_state = env.reset(rng_reset, env_params)
        rng_control, rng = crandom.split(rng)
        control_params = controller.reset(env_state, env_params, controller.init_control_params, rng_control)
        errs = []
        for _ in range(T):  # run_one_step, :520-538
            rng, rng_act, rng_step, rng_control = crandom.split(rng, 4)
            action, control_params, control_info = controller(obs, env_state, env_params, rng_act, control_params, info)
            if hasattr(action, "detach"):
                action = action.detach().cpu().numpy()
            obs, env_state, reward, done, info = env.step(rng_step, env_state, action, env_params)
            rng, rng_control = crandom.split(rng)
            errs.append(info["err_pos"])
        return rng, np.asarray(errs)

    t0 = time_module.time()
    num_eps = int(total_steps // T)
    err_pos_ep = []
    rng, rng_reset_meta = crandom.split(rng)
    for i, rng_reset in enumerate(crandom.split(rng_reset_meta, num_trajs)):
        if verbose:
            print(f"[DEBUG] test traj {i+1}")
        for _ in range(max(num_eps // num_trajs, 1)):
            rng, err_pos = run_one_ep(rng_reset, rng)
            err_pos_ep.append(err_pos.mean())
    err_pos_ep = np.asarray(err_pos_ep)
    pos_mean, pos_std = err_pos_ep.mean(), err_pos_ep.std()
    if verbose:
        print(f"env running time: {time_module.time()-t0:.2f}s")
        print(f"err_pos mean: {pos_mean:.3f}, std: {pos_std:.3f}")
        print(f"${pos_mean*100:.2f} \\pm {pos_std*100:.2f}$")
    if save:
        from .. import get_package_path
        save_path = f"{get_package_path()}/../results"
        os.makedirs(save_path, exist_ok=True)
        with open(f"{save_path}/eval_err_pos_{filename}.pkl", "wb") as f:
            pickle.dump(np.array(err_pos_ep), f)
    return err_pos_ep


def get_controller(env, controller_name, controller_params=None, debug=False, device=None, process_group=None,
                   compute_info=True):
    """quadrotor.py:670-752."""
    import torch

    def parse_sample_params(param_text):
        if not param_text:
            return 8192, 32, 0.01, 0.5
        parts = param_text.split("_")
        return int(parts[0][1:]), int(parts[1][1:]), float(parts[2][3:]), 0.5

    device = device if device is not None else (env.device if env.device is not None else "cuda")

    def get_sample_mean(H):
        dp = env.default_params
        th = (dp.m * dp.g / dp.max_thrust) * 2.0 - 1.0
        return torch.tensor([th, 0.0, 0.0, 0.0], dtype=torch.float32, device=device).repeat(H, 1)

    if controller_name == "pid":
        control_params = controllers.PIDParams(Kp=10.0, Kd=5.0, Ki=0.0, Kp_att=10.0)
        return controllers.PIDController(env, control_params=control_params), control_params
    if controller_name == "random":
        return controllers.RandomController(env, None), None
    if controller_name == "mppi":
        N, H, lam, sigma = parse_sample_params(controller_params)
        a_cov = (torch.eye(env.action_dim, dtype=torch.float32, device=device) * sigma ** 2).repeat(H, 1, 1)
        control_params = controllers.MPPIParams(gamma_mean=1.0, gamma_sigma=0.0, discount=1.0, sample_sigma=sigma,
                                                a_mean=get_sample_mean(H), a_cov=a_cov)
        return controllers.MPPIController(env=env, control_params=control_params, N=N, H=H, lam=lam, device=device,
                                          process_group=process_group, compute_info=compute_info), control_params
    if "covo" in controller_name:
        N, H, lam, sigma = parse_sample_params(controller_params)
        mode = "offline" if "offline" in controller_name else "online"
        if "online" not in controller_name and "offline" not in controller_name:
            print("[DEBUG] unset mode, CoVO mode set to online")
        control_params = controllers.CoVOParams(
            gamma_mean=1.0, gamma_sigma=0.0, discount=1.0, sample_sigma=sigma, a_mean=get_sample_mean(H),
            a_cov=torch.eye(H * env.action_dim, dtype=torch.float32, device=device) * sigma ** 2,
            a_cov_offline=torch.zeros((H, env.action_dim, env.action_dim), dtype=torch.float32, device=device))
        return controllers.CoVOController(env=env, control_params=control_params, N=N, H=H, lam=lam, mode=mode,
                                          device=device, process_group=process_group,
                                          compute_info=compute_info), control_params
    raise NotImplementedError(controller_name)


@pydataclass
class Args:
    """quadrotor.py:755-766."""
    task: str = "tracking"
    controller: str = "lqr"
    controller_params: str = ""
    obs_type: str = "quad"
    debug: bool = False
    mode: str = "render"
    lower_controller: str = "base"
    noDR: bool = False
    disturb_type: str = "gaussian"
    name: str = ""
    host_env: bool = False  # (not in quadjax) eval with the Python env step instead of the device one


def main(args: Args):
    """quadrotor.py:769-803 (eval mode; `render` needs matplotlib/meshcat post-processing, out of scope)."""
    env = Quad3D(task=args.task, obs_type=args.obs_type, lower_controller=args.lower_controller,
                 enable_randomizer=not args.noDR, disturb_type=args.disturb_type, disable_rollover_terminate=True,
                 generate_noisy_state=True, device="cuda")
    print("starting test...")
    # eval never reads the controller's info dict: quadjax's jitted run_one_step drops pos_mean / pos_std as dead code
    # (quadrotor.py:523-538), here the per-step position statistics are simply not requested
    controller, control_params = get_controller(env, args.controller, args.controller_params, compute_info=args.mode != "eval")
    if args.mode == "eval":
        if not args.host_env and hasattr(controller, "core"):
            # same protocol, env step on the device, whole episodes enqueued by one C call (eval_env_device)
            return eval_env_device(env, controller=controller, total_steps=300 * 4 * 10)
        return eval_env(env, controller=controller, total_steps=300 * 4 * 10, filename=args.name)
    raise NotImplementedError(args.mode)


def _cli():
    ap = argparse.ArgumentParser(description="quadjax-compatible driver (same flag names as quadrotor.py:755-766)")
    for f, default in Args().__dict__.items():
        if isinstance(default, bool):
            ap.add_argument(f"--{f}", action="store_true")
        else:
            ap.add_argument(f"--{f}", type=type(default), default=default)
    main(Args(**vars(ap.parse_args())))


if __name__ == "__main__":
    _cli()
