"""Host-side reward functions and reference-trajectory generators (numpy).

Mirrors quadjax/dynamics/utils.py: rewards 266-294, generators 49-53 / 87-180 / 183-251.  These
feed the environment plumbing (reset, info, episode reward); the per-sample rollout reward is the
HIP kernel's (csrc/quad_model.hpp).  Random draws come from covo_mpc_amd.random keys (jax's
threefry stream is unpinned), the draw structure of each generator is kept.
"""
import numpy as np

from .. import random as crandom


def log_pos_fn(err_pos):
    """utils.py:266-274."""
    l = np.log(err_pos + 1)
    return (err_pos * 0.4 + np.clip(l * 4, 0, 1) * 0.4 + np.clip(l * 8, 0, 1) * 0.2
            + np.clip(l * 16, 0, 1) * 0.1 + np.clip(l * 32, 0, 1) * 0.1)


def tracking_penyaw_reward_fn(state, params=None):
    """utils.py:285-294."""
    err_pos = np.linalg.norm(state.pos_tar - state.pos)
    err_vel = np.linalg.norm(state.vel_tar - state.vel)
    q = state.quat
    yaw = np.arctan2(2 * (q[3] * q[2] + q[0] * q[1]), 1 - 2 * (q[1] ** 2 + q[2] ** 2))
    return np.float32(1.3 - 0.05 * err_vel - log_pos_fn(err_pos) - np.abs(yaw) * 0.2)


def tracking_realworld_reward_fn(state, params=None):
    """utils.py:297-313."""
    pos_err = np.mean((state.pos - state.pos_tar) ** 2)
    quat_err = 1 - state.quat[3] ** 2
    return np.float32(-(5.0 * pos_err + 3.0 * quat_err) * 0.02)


def generate_fixed_traj(max_steps, dt, key):
    """utils.py:49-53."""
    z = np.zeros((max_steps, 3), dtype=np.float32)
    return z, z.copy(), z.copy()


def _lissa(max_steps, dt, key, f1, f2):
    key_amp, key_phase = crandom.split(key, 2)
    amp = crandom.uniform(key_amp, (3, 2), -1.0, 1.0, np.float64)
    phase = crandom.uniform(key_phase, (3, 2), -np.pi, np.pi, np.float64)
    ts = np.arange(0, max_steps + 50) * dt
    w1, w2 = 2 * np.pi * f1, 2 * np.pi * f2
    pos = np.stack([amp[i, 0] * np.sin(w1 * ts + phase[i, 0]) + amp[i, 1] * np.sin(w2 * ts + phase[i, 1])
                    for i in range(3)], axis=1)
    pos = pos - pos[0]
    vel = np.stack([amp[i, 0] * w1 * np.cos(w1 * ts + phase[i, 0]) + amp[i, 1] * w2 * np.cos(w2 * ts + phase[i, 1])
                    for i in range(3)], axis=1)
    acc = np.stack([-amp[i, 0] * w1 ** 2 * np.sin(w1 * ts + phase[i, 0])
                    - amp[i, 1] * w2 ** 2 * np.sin(w2 * ts + phase[i, 1]) for i in range(3)], axis=1)
    return pos.astype(np.float32), vel.astype(np.float32), acc.astype(np.float32)


def generate_lissa_traj(max_steps, dt, key):
    """utils.py:87-130."""
    return _lissa(max_steps, dt, key, 0.2, 0.4)


def generate_lissa_traj_slow(max_steps, dt, key):
    """utils.py:133-180."""
    return _lissa(max_steps, dt, key, 0.1, 0.1)


def generate_zigzag_traj(max_steps, dt, key):
    """utils.py:183-251: 8 segments x 40 points; quirks kept (shared key arrays for key-points and
    angles :187-188, segments 0 and 1 both use keys[1] :238-241, distance U(1,1.5) :219, velocity
    divides by point_per_seg+1 :231-236)."""
    point_per_seg = 40
    num_seg = max_steps // point_per_seg + 1
    keys = crandom.split(key, num_seg + 1)
    prev = crandom.uniform(keys[0], (3,), -1.0, 1.0, np.float64)
    prev = prev / np.linalg.norm(prev) * 0.1
    pos_segs, vel_segs = [], []
    k = keys[1]
    for i in range(num_seg):
        to_center = -prev / np.linalg.norm(prev)
        d_theta, d_phi = crandom.uniform(k, (2,), -np.pi / 3, np.pi / 3, np.float64)
        theta = np.arccos(to_center[2]) + d_theta
        phi = np.arctan2(to_center[1], to_center[0]) + d_phi
        direction = np.array([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)])
        distance = 1.0 + 0.5 * float(crandom.uniform(k, (1,), 0.0, 1.0, np.float64)[0])
        nxt = prev + distance * direction
        pos_segs.append(np.stack([np.linspace(a, b, point_per_seg, endpoint=False) for a, b in zip(prev, nxt)], -1))
        vel_segs.append((nxt - prev) / (point_per_seg + 1) * np.ones((point_per_seg, 3)) / dt)
        prev = nxt
        k = keys[i + 1]
    pos = np.concatenate(pos_segs, 0)
    pos = pos - pos[0]
    vel = np.concatenate(vel_segs, 0)
    return pos.astype(np.float32), vel.astype(np.float32), np.zeros_like(pos, dtype=np.float32)
