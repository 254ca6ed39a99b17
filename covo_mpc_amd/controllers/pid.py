"""Geometric PD position + P attitude controller: quadjax/controllers/pid.py:11-84.

Host-side numpy (tiny serial math, SURVEY.md component #4): a standalone baseline and the
nominal-trajectory generator of covo-offline's per-episode Sigma table (covo.py:48-56).
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field

import numpy as np

from ..dynamics import geom
from .base import BaseController


def _arr(x):
    return field(default_factory=lambda: np.asarray(x, dtype=np.float32))


@dataclass(frozen=True)
class PIDParams:
    """pid.py:11-22."""
    Kp: float = 4.0
    Kd: float = 4.0
    Ki: float = 1.0
    Kp_att: float = 4.0
    Ki_att: float = 1.0
    integral: np.ndarray = _arr([0.0, 0.0, 0.0])
    quat_desired: np.ndarray = _arr([0.0, 0.0, 0.0, 1.0])
    att_integral: np.ndarray = _arr([0.0, 0.0, 0.0])

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


class PIDController(BaseController):
    def __init__(self, env, control_params) -> None:
        super().__init__(env, control_params)
        self.param = self.env.default_params  # pid.py:33 -- DEFAULT m, g even under DR

    def __call__(self, obs, state, env_param, rng_act, control_params, info=None):
        f32 = np.float32
        p = self.param
        Q = geom.qtoQ(state.quat)
        f_d = f32(p.m) * (np.array([0.0, 0.0, p.g], dtype=f32) - f32(control_params.Kp) * (state.pos - state.pos_tar)
                          - f32(control_params.Kd) * (state.vel - state.vel_tar)
                          - f32(control_params.Ki) * control_params.integral + state.acc_tar)  # pid.py:44-50
        thrust = np.clip((Q.T @ f_d)[2], 0.0, p.max_thrust)  # :51-52
        f_d_norm = np.linalg.norm(f_d)
        f_d_norm = f32(1e-3) if f_d_norm < 1e-3 else f_d_norm  # :56-57
        z_d = f_d / f_d_norm
        axis_angle = np.cross(np.array([0.0, 0.0, 1.0], dtype=f32), z_d)
        angle = np.linalg.norm(axis_angle)
        angle = f32(5e-4) if angle < 1e-3 else angle  # :61
        axis = np.array([0.0, 0.0, 1.0], dtype=f32) if angle < 1e-3 else axis_angle / angle  # :62
        R_d = geom.axisangletoR(axis.astype(f32), angle)
        quat_desired = geom.Qtoq(R_d)
        R_e = R_d.T @ Q
        angle_err = geom.vee(R_e - R_e.T)
        omega_d = -f32(control_params.Kp_att) * angle_err  # :68
        action = np.concatenate([[thrust / p.max_thrust * 2.0 - 1.0], omega_d / p.max_omega]).astype(f32)  # :71-76
        integral = control_params.integral + (state.pos - state.pos_tar) * f32(env_param.dt)  # :79
        return action, control_params.replace(quat_desired=quat_desired.astype(f32), integral=integral.astype(f32)), None
