"""PD position loop + proportional attitude loop on SO(3), emitting (thrust, body-rate) actions.

Behavioural twin of quadjax/controllers/pid.py:11-84 -- same parameter names, call signature and returned triple -- used
in two places: as the stand-alone `--controller pid` baseline and as covo-offline's expansion controller, whose
PID-tracked trajectory supplies the nominal means of the per-episode Sigma table (covo.py:48-99; on the device that loop is
csrc/pid_nominal.hip, which tests compare against this host version).  Everything here is a handful of 3-vectors per call,
so it is plain numpy in fp32 (the reference's dtype).

The law, in the order it is evaluated below (reference lines in parentheses):
  1. wanted force   F = m (g e3 - Kp e_p - Kd e_v - Ki I + a_ref),  e_p = p - p_ref, e_v = v - v_ref, I = integral  (44-50)
  2. thrust         = clip(F . body_z, 0, max_thrust), body_z = third column of the attitude matrix                  (51-52)
  3. wanted attitude: the rotation about e3 x z_d that tilts e3 onto z_d = F/|F|.  The reference feeds |e3 x z_d|
     -- the SINE of the tilt -- to the axis-angle map as if it were the angle, and replaces tilts below 1e-3 by a 5e-4
     tilt about e3; both quirks are kept, the nominal trajectories depend on them                                    (55-63)
  4. body rates     = -Kp_att vee(R_e - R_e^T), R_e = R_d^T R                                                        (65-68)
  5. action         = [2 thrust/max_thrust - 1, rates / max_omega]; the position integral advances by e_p dt          (71-82)
The gains act on the env's DEFAULT parameters (m, g, limits), also under domain randomisation (pid.py:33).
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field

import numpy as np

from ..dynamics import geom
from .base import BaseController

_F32 = np.float32
_E3 = np.array([0.0, 0.0, 1.0], dtype=_F32)


def _vec(*values):
    return field(default_factory=lambda: np.array(values, dtype=_F32))


@dataclass(frozen=True)
class PIDParams:
    """Gains and controller state (pid.py:11-22); `replace` mirrors flax.struct's."""
    Kp: float = 4.0
    Kd: float = 4.0
    Ki: float = 1.0
    Kp_att: float = 4.0
    Ki_att: float = 1.0
    integral: np.ndarray = _vec(0.0, 0.0, 0.0)
    quat_desired: np.ndarray = _vec(0.0, 0.0, 0.0, 1.0)
    att_integral: np.ndarray = _vec(0.0, 0.0, 0.0)

    def replace(self, **changes):
        return dataclasses.replace(self, **changes)


def wanted_force(gains: PIDParams, state, mass, gravity):
    """Step 1: the force that cancels gravity, follows the reference acceleration and pulls the errors to zero."""
    pull = (_F32(gains.Kp) * (state.pos - state.pos_tar) + _F32(gains.Kd) * (state.vel - state.vel_tar)
            + _F32(gains.Ki) * gains.integral)
    return _F32(mass) * (_F32(gravity) * _E3 - pull + state.acc_tar)


def tilt_towards(force):
    """Step 3: rotation matrix whose z axis is tilted from e3 towards `force`, with the reference's guards and its
    sine-for-angle convention."""
    size = np.linalg.norm(force)
    z_d = force / (_F32(1e-3) if size < 1e-3 else size)
    rot_vec = np.cross(_E3, z_d)
    tilt = np.linalg.norm(rot_vec)
    if tilt < 1e-3:
        # pid.py:61 replaces the angle by 5e-4 first, so the `< 1e-3` test of :62 is then always true: axis = e3
        return geom.axisangletoR(_E3.copy(), _F32(5e-4))
    return geom.axisangletoR((rot_vec / tilt).astype(_F32), tilt)


class PIDController(BaseController):
    def __init__(self, env, control_params) -> None:
        super().__init__(env, control_params)
        self.param = env.default_params  # nominal model, also when the env's parameters are randomised

    def __call__(self, obs, state, env_param, rng_act, control_params, info=None):
        model, gains = self.param, control_params
        attitude = geom.qtoQ(state.quat)
        force = wanted_force(gains, state, model.m, model.g)
        thrust = min(max(float(attitude[:, 2] @ force), 0.0), float(model.max_thrust))
        target = tilt_towards(force)
        mismatch = target.T @ attitude
        rates = -_F32(gains.Kp_att) * geom.vee(mismatch - mismatch.T)
        action = np.empty(4, dtype=_F32)
        action[0] = thrust / model.max_thrust * 2.0 - 1.0
        action[1:] = rates / model.max_omega
        drift = gains.integral + (state.pos - state.pos_tar) * _F32(env_param.dt)
        return action, gains.replace(quat_desired=geom.Qtoq(target).astype(_F32), integral=drift.astype(_F32)), None
