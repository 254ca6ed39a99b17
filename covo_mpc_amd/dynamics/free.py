"""Host-side single-state quadrotor step (numpy float32): quadjax/dynamics/free.py.

This is environment plumbing (one state per control step); the N x H rollout of the
same model is the HIP kernel csrc/rollout.hip.  Disturbance models free.py:9-72.
"""
import numpy as np

from .. import random as crandom
from .dataclass import Action3D, EnvParams3D, EnvState3D


def get_quadrotor_1st_order_dyn(disturb_type: str = "periodic"):
    """free.py:8-204: returns (step_fn, dynamics_fn)."""
    f32 = np.float32

    def period_disturb(key, params, state):  # free.py:10-24
        if state.time % params.disturb_period == 0:
            return crandom.uniform(key, (3,), -params.disturb_scale, params.disturb_scale)
        return state.f_disturb

    def sin_disturb(key, params, state):  # free.py:27-38
        dp = np.asarray(params.disturb_params, dtype=f32)
        scale = dp[:3] * f32(params.disturb_scale)
        period = dp[:3] * f32(params.disturb_period / 3) + f32(params.disturb_period)
        phase = dp[3:6] * f32(2 * np.pi)
        return (scale * np.sin(f32(2 * np.pi) / period * f32(state.time) + phase)).astype(f32)

    def drag_disturb(key, params, state):  # free.py:41-47
        rel = state.vel - np.asarray(params.disturb_params, dtype=f32)[:3] * f32(0.5)
        return (-abs(f32(params.disturb_scale)) * rel * np.abs(rel) / f32(1.5 ** 2)).astype(f32)

    def mixed_disturb(key, params, state):  # free.py:50-56
        return ((drag_disturb(key, params, state) + sin_disturb(key, params, state)
                 + period_disturb(key, params, state)) / f32(3)).astype(f32)

    table = {
        "periodic": period_disturb, "sin": sin_disturb, "drag": drag_disturb, "mixed": mixed_disturb,
        "gaussian": lambda key, params, state: (f32(params.dyn_noise_scale) * crandom.normal(key, (3,))).astype(f32),
        "none": lambda key, params, state: np.zeros(3, dtype=f32),
    }
    if disturb_type not in table:
        raise NotImplementedError(disturb_type)
    disturb_func = table[disturb_type]

    def quad_dynamics_bodyrate(x, u, params: EnvParams3D, dt, key=None):  # free.py:74-112
        x = np.asarray(x, dtype=f32)
        u = np.asarray(u, dtype=f32) * f32(params.action_scale)
        thrust, omega_tar = u[0], u[1:4]
        r, q, v, om, f = x[0:3], x[3:7] / np.linalg.norm(x[3:7]), x[7:10], x[10:13], x[13:16]
        qx, qy, qz, qw = q
        Qz = np.array([2 * (qx * qz + qy * qw), 2 * (qy * qz - qx * qw), qw * qw - qx * qx - qy * qy + qz * qz], dtype=f32)
        q_dot = f32(0.5) * np.concatenate([qw * om + np.cross(q[:3], om), [-np.dot(q[:3], om)]]).astype(f32)
        v_dot = np.array([0, 0, -params.g], dtype=f32) + f32(1.0) / f32(params.m) * (Qz * thrust + f)
        dt = f32(dt)
        a = f32(params.alpha_bodyrate)
        return np.concatenate([r + v * dt, q + q_dot * dt, v + v_dot * dt, a * om + (f32(1) - a) * omega_tar, f]).astype(f32)

    def free_dynamics_3d_bodyrate(env_params, env_state: EnvState3D, env_action: Action3D, key, sim_dt):  # free.py:114-202
        omega_tar = (env_action.torque / env_params.max_torque * env_params.max_omega).astype(f32)
        u = np.concatenate([[env_action.thrust], omega_tar]).astype(f32)
        x = np.concatenate([env_state.pos, env_state.quat, env_state.vel, env_state.omega, env_state.f_disturb]).astype(f32)
        key, key_dyn = crandom.split(key)
        x_new = quad_dynamics_bodyrate(x, u, env_params, sim_dt, key_dyn)
        quat = x_new[3:7] / np.linalg.norm(x_new[3:7])
        disturb_key, key = crandom.split(key)
        f_disturb = np.asarray(disturb_func(disturb_key, env_params, env_state), dtype=f32)
        time = env_state.time + 1
        idx = min(max(time, 0), env_state.pos_traj.shape[0] - 1)  # JAX gather clamps
        action = np.concatenate([[env_action.thrust / env_params.max_thrust * 2.0 - 1.0],
                                 env_action.torque / env_params.max_torque]).astype(f32)
        return env_state.replace(
            pos=x_new[0:3], vel=x_new[7:10], omega=x_new[10:13], quat=quat.astype(f32),
            pos_tar=env_state.pos_traj[idx], vel_tar=env_state.vel_traj[idx], acc_tar=env_state.acc_traj[idx],
            omega_tar=omega_tar, last_thrust=float(env_action.thrust), last_torque=env_action.torque, time=time,
            f_disturb=f_disturb,
            vel_hist=np.concatenate([env_state.vel_hist[1:], env_state.vel[None]]),
            omega_hist=np.concatenate([env_state.omega_hist[1:], env_state.omega[None]]),
            action_hist=np.concatenate([env_state.action_hist[1:], action[None]]),
        )

    return free_dynamics_3d_bodyrate, quad_dynamics_bodyrate
