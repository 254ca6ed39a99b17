"""State / parameter containers mirroring quadjax/dynamics/dataclass.py.

`EnvParams3D` (dataclass.py:40-100) and `EnvState3D` (dataclass.py:10-37) are immutable with
`.replace(...)` like flax.struct dataclasses.  Host fields are numpy (the env step is host
plumbing); `DeviceState` is the HBM image the controllers hand to the HIP kernels
(include/covo_hip.h "Data layouts": state float[32] + trajectory float[T][3]).
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import Any, Optional

import numpy as np

from .._lib import COVO_STATE_FLOATS, EnvParamsC

ST_POS, ST_VEL, ST_QUAT, ST_OMEGA, ST_FDIST, ST_POSTAR, ST_VELTAR, ST_ACCTAR, ST_TIME = 0, 3, 6, 10, 13, 16, 19, 22, 25


def _arr(x):
    return field(default_factory=lambda: np.asarray(x, dtype=np.float32))


@dataclass(frozen=True)
class EnvParams3D:
    """quadjax/dynamics/dataclass.py:40-100 (same names and defaults)."""
    max_speed: float = 8.0
    max_torque: np.ndarray = _arr([9e-3, 9e-3, 2e-3])
    max_omega: np.ndarray = _arr([10.0, 10.0, 3.0])
    max_thrust: float = 0.8
    dt: float = 0.02
    g: float = 9.81
    m: float = 0.027
    m_mean: float = 0.027
    m_std: float = 0.003
    I: np.ndarray = _arr([[1.7e-5, 0.0, 0.0], [0.0, 1.7e-5, 0.0], [0.0, 0.0, 3.0e-5]])
    I_diag_mean: np.ndarray = _arr([1.7e-5, 1.7e-5, 3.0e-5])
    I_diag_std: np.ndarray = _arr([0.2e-5, 0.2e-5, 0.3e-5])
    l: float = 0.3
    l_mean: float = 0.3
    l_std: float = 0.1
    hook_offset: np.ndarray = _arr([0.0, 0.0, -0.01])
    hook_offset_mean: np.ndarray = _arr([0.0, 0.0, -0.02])
    hook_offset_std: np.ndarray = _arr([0.01, 0.01, 0.01])
    action_scale: float = 1.0
    action_scale_mean: float = 1.0
    action_scale_std: float = 0.1
    alpha_bodyrate: float = 0.5
    alpha_thrust: float = 0.6
    alpha_bodyrate_mean: float = 0.5
    alpha_bodyrate_std: float = 0.1
    max_steps_in_episode: int = 300
    rope_taut_therehold: float = 1e-4
    traj_obs_len: int = 5
    traj_obs_gap: int = 5
    d_offset: np.ndarray = _arr([0.0] * 6)
    disturb_period: int = 50
    disturb_scale: float = 0.2
    disturb_params: np.ndarray = _arr([0.0] * 6)
    curri_params: float = 1.0
    adapt_horizon: int = 4
    dyn_noise_scale: float = 0.05
    obs_noise_scale: float = 0.05

    def replace(self, **kw) -> "EnvParams3D":
        return dataclasses.replace(self, **kw)

    def to_c(self, rollover_terminate: bool = False, reward: str = "penyaw", disturb_type: str = "none",
             reset_task=None) -> EnvParamsC:
        """The rollout-relevant subset as struct covo_env_params.  `rollover_terminate` (the env's `not
        disable_rollover_terminate`, envs/quadrotor.py:486-490), `reward` (which function env.reward_fn is,
        quadrotor.py:49-84) and `disturb_type` (quadrotor.py:35,87-89) are attributes of Quad3D, not of the parameters.
        `reset_task`: the Quad3D task whose trajectory generator the DEVICE env's auto-reset runs (base.py:22-40); None = no
        auto-reset (only covo_env_step* / covo_run_episode* read it)."""
        from .._lib import DISTURB_KINDS, REWARD_KINDS, TRAJ_KINDS
        c = EnvParamsC()
        c.reset_traj, c.reserved0 = TRAJ_KINDS[reset_task], 0
        c.reset_dt, c.reset_disturb_scale = float(self.dt), float(self.disturb_scale)
        c.reward_kind, c.disturb_kind = REWARD_KINDS[reward], DISTURB_KINDS[disturb_type]
        c.disturb_period, c.disturb_scale = int(self.disturb_period), float(self.disturb_scale)
        for i in range(6):
            c.disturb_params[i] = float(self.disturb_params[i])
        c.dyn_noise_scale = float(self.dyn_noise_scale)
        c.max_thrust = float(self.max_thrust)
        for i in range(3):
            c.max_torque[i] = float(self.max_torque[i])
            c.max_omega[i] = float(self.max_omega[i])
        c.dt, c.g, c.m = float(self.dt), float(self.g), float(self.m)
        c.action_scale, c.alpha_bodyrate = float(self.action_scale), float(self.alpha_bodyrate)
        c.max_steps_in_episode = int(self.max_steps_in_episode)
        c.pos_limit = 3.0  # envs/quadrotor.py:484
        c.rollover_terminate = 1 if rollover_terminate else 0
        return c


@dataclass(frozen=True)
class Action3D:
    """quadjax/dynamics/dataclass.py:103-106."""
    thrust: float
    torque: np.ndarray


@dataclass(frozen=True)
class DeviceState:
    """HBM image of the state the MPC path reads.  packed: float32[32]; trajectories float32[T,3]."""
    packed: Any
    pos_traj: Any
    vel_traj: Any
    time: Optional[int] = None  # host copy of state.time when known (lets table look-ups be plain views)

    @property
    def T(self) -> int:
        return int(self.pos_traj.shape[0])


@dataclass(frozen=True)
class EnvState3D:
    """quadjax/dynamics/dataclass.py:10-37 (host copy, numpy float32)."""
    pos: np.ndarray
    vel: np.ndarray
    quat: np.ndarray
    omega: np.ndarray
    omega_tar: np.ndarray
    pos_traj: np.ndarray
    vel_traj: np.ndarray
    acc_traj: np.ndarray
    pos_tar: np.ndarray
    vel_tar: np.ndarray
    acc_tar: np.ndarray
    last_thrust: float
    last_torque: np.ndarray
    time: int
    f_disturb: np.ndarray
    vel_hist: np.ndarray
    omega_hist: np.ndarray
    action_hist: np.ndarray
    control_params: Any = 0.0
    traj_dev: Optional[tuple] = None  # (pos_traj, vel_traj) device tensors, uploaded once per episode

    def replace(self, **kw) -> "EnvState3D":
        return dataclasses.replace(self, **kw)

    def pack(self) -> np.ndarray:
        x = np.zeros(COVO_STATE_FLOATS, dtype=np.float32)
        x[ST_POS:ST_POS + 3] = self.pos
        x[ST_VEL:ST_VEL + 3] = self.vel
        x[ST_QUAT:ST_QUAT + 4] = self.quat
        x[ST_OMEGA:ST_OMEGA + 3] = self.omega
        x[ST_FDIST:ST_FDIST + 3] = self.f_disturb
        x[ST_POSTAR:ST_POSTAR + 3] = self.pos_tar
        x[ST_VELTAR:ST_VELTAR + 3] = self.vel_tar
        x[ST_ACCTAR:ST_ACCTAR + 3] = self.acc_tar
        x[ST_TIME:ST_TIME + 1] = np.asarray([self.time], dtype=np.int32).view(np.float32)
        return x

    def to_device(self, device) -> DeviceState:
        import torch
        if self.traj_dev is not None and self.traj_dev[0].device == torch.device(device):
            pt, vt = self.traj_dev
        else:
            pt = torch.from_numpy(np.ascontiguousarray(self.pos_traj, dtype=np.float32)).to(device)
            vt = torch.from_numpy(np.ascontiguousarray(self.vel_traj, dtype=np.float32)).to(device)
        packed = torch.from_numpy(self.pack()).to(device, non_blocking=True)
        return DeviceState(packed=packed, pos_traj=pt, vel_traj=vt, time=int(self.time))


def as_device_state(state, device) -> DeviceState:
    if isinstance(state, DeviceState):
        return state
    return state.to_device(device)
