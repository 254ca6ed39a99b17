"""numpy restatement of the quadjax MPC control step.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the JAX reference cannot run here (see ``oracle/__init__``).

Every function cites the reference lines (relative to /root/reference/quadjax)
it restates.  All arithmetic is done in ``dtype`` (np.float32 mirrors JAX's
default; np.float64 is the accuracy yardstick the HIP kernels are compared
against).  State arrays may carry leading batch axes (..., 3) so that one call
evaluates all N samples of a rollout step -- the same thing ``jax.vmap`` does at
controllers/covo.py:230-232.

Randomness is never drawn here implicitly: epsilon, disturbance vectors and
observation noise are explicit inputs (jax's threefry bitstream is unpinned and
unavailable).  The trajectory generators take a ``numpy.random.Generator``.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import Optional

import numpy as np


# ----------------------------------------------------------------------------
# constants: dynamics/dataclass.py:40-100 (EnvParams3D defaults)
# ----------------------------------------------------------------------------
@dataclass
class Params:
    max_speed: float = 8.0
    max_torque: tuple = (9e-3, 9e-3, 2e-3)
    max_omega: tuple = (10.0, 10.0, 3.0)
    max_thrust: float = 0.8
    dt: float = 0.02
    g: float = 9.81
    m: float = 0.027
    m_mean: float = 0.027
    m_std: float = 0.003
    I_diag_mean: tuple = (1.7e-5, 1.7e-5, 3.0e-5)
    I_diag_std: tuple = (0.2e-5, 0.2e-5, 0.3e-5)
    action_scale: float = 1.0
    action_scale_mean: float = 1.0
    action_scale_std: float = 0.1
    alpha_bodyrate: float = 0.5
    alpha_bodyrate_mean: float = 0.5
    alpha_bodyrate_std: float = 0.1
    max_steps_in_episode: int = 300
    traj_obs_len: int = 5
    traj_obs_gap: int = 5
    disturb_period: int = 50
    disturb_scale: float = 0.2
    disturb_params: tuple = (0.0,) * 6
    adapt_horizon: int = 4
    dyn_noise_scale: float = 0.05
    obs_noise_scale: float = 0.05

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)

    def fp32(self):
        """Every real parameter rounded to float32 -- the reference's parameters ARE fp32 numbers (JAX
        default dtype), so fp64 yardstick runs must start from the same rounded constants."""
        kw = {}
        for f in dataclasses.fields(self):
            v = getattr(self, f.name)
            if isinstance(v, float):
                kw[f.name] = float(np.float32(v))
            elif isinstance(v, tuple):
                kw[f.name] = tuple(float(np.float32(x)) for x in v)
        return dataclasses.replace(self, **kw)


@dataclass
class State:
    """dynamics/dataclass.py:10-37 restricted to what the MPC path reads."""
    pos: np.ndarray
    vel: np.ndarray
    quat: np.ndarray  # (x, y, z, w)
    omega: np.ndarray
    f_disturb: np.ndarray
    pos_tar: np.ndarray
    vel_tar: np.ndarray
    acc_tar: np.ndarray
    time: int
    pos_traj: np.ndarray  # (T, 3), shared by all samples
    vel_traj: np.ndarray
    acc_traj: np.ndarray

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)

    def astype(self, dtype):
        kw = {}
        for f in dataclasses.fields(self):
            v = getattr(self, f.name)
            kw[f.name] = v if f.name == "time" else np.asarray(v, dtype=dtype)
        return State(**kw)


def _c(x, dtype):
    return np.asarray(x, dtype=dtype)


def norm(v):
    """jnp.linalg.norm on the last axis (sqrt of sum of squares)."""
    return np.sqrt(np.sum(v * v, axis=-1))


# ----------------------------------------------------------------------------
# geometry: dynamics/geom.py
# ----------------------------------------------------------------------------
def hat(v):
    """geom.py:35-39."""
    z = np.zeros_like(v[..., 0])
    return np.stack(
        [
            np.stack([z, -v[..., 2], v[..., 1]], -1),
            np.stack([v[..., 2], z, -v[..., 0]], -1),
            np.stack([-v[..., 1], v[..., 0], z], -1),
        ],
        -2,
    )


def geom_L(q):
    """geom.py:41-53: L(q) = [[s I + hat(v), v], [-v^T, s]]."""
    s = q[..., 3]
    v = q[..., :3]
    eye = np.eye(3, dtype=q.dtype)
    left_up = s[..., None, None] * eye + hat(v)  # (...,3,3)
    left = np.concatenate([left_up, -v[..., None, :]], axis=-2)  # (...,4,3)
    right = np.concatenate([v, s[..., None]], axis=-1)[..., :, None]  # (...,4,1)
    return np.concatenate([left, right], axis=-1)  # (...,4,4)


def geom_H(dtype):
    """geom.py:55."""
    return np.vstack([np.eye(3, dtype=dtype), np.zeros((1, 3), dtype=dtype)])


def qtoQ(q):
    """geom.py:68-77: H^T T L(q) T L(q) H, literally (4x4 matmuls)."""
    T = np.diag(np.asarray([-1, -1, -1, 1], dtype=q.dtype))
    H = geom_H(q.dtype)
    Lq = geom_L(q)
    return H.T @ T @ Lq @ T @ Lq @ H


def Qtoq(Q):
    """geom.py:79-87."""
    tr = 1 + Q[0, 0] + Q[1, 1] + Q[2, 2]
    w = 0.5 * np.sqrt(tr)
    xyz = 0.5 / np.sqrt(tr) * np.array(
        [Q[2, 1] - Q[1, 2], Q[0, 2] - Q[2, 0], Q[1, 0] - Q[0, 1]], dtype=Q.dtype
    )
    return np.concatenate([xyz, [w]]).astype(Q.dtype)


def axisangletoR(axis, angle):
    """geom.py:106-112."""
    axis = axis / norm(axis)
    K = hat(axis)
    return np.eye(3, dtype=axis.dtype) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def vee(R):
    """geom.py:114-120."""
    return np.array([R[2, 1], R[0, 2], R[1, 0]], dtype=R.dtype)


# ----------------------------------------------------------------------------
# reward: dynamics/utils.py:266-294
# ----------------------------------------------------------------------------
def log_pos_fn(err_pos):
    """utils.py:266-274."""
    l = np.log(err_pos + 1)
    return (
        err_pos * 0.4
        + np.clip(l * 4, 0, 1) * 0.4
        + np.clip(l * 8, 0, 1) * 0.2
        + np.clip(l * 16, 0, 1) * 0.1
        + np.clip(l * 32, 0, 1) * 0.1
    )


def tracking_penyaw_reward_fn(state: State):
    """utils.py:285-294 (yaw is taken from the STORED quaternion, un-normalised)."""
    dt = state.pos.dtype.type
    err_pos = norm(state.pos_tar - state.pos)
    err_vel = norm(state.vel_tar - state.vel)
    q = state.quat
    yaw = np.arctan2(
        2 * (q[..., 3] * q[..., 2] + q[..., 0] * q[..., 1]),
        1 - 2 * (q[..., 1] ** 2 + q[..., 2] ** 2),
    )
    return dt(1.3) - dt(0.05) * err_vel - log_pos_fn(err_pos) - np.abs(yaw) * dt(0.2)


def tracking_realworld_reward_fn(state: State):
    """utils.py:297-313 (task "tracking_slow", envs/quadrotor.py:58-66): quadratic position cost + attitude cost on
    the STORED quaternion's w."""
    dt = state.pos.dtype.type
    pos_err = state.pos - state.pos_tar
    pos_err = np.mean(pos_err ** 2, axis=-1)
    quat_err = dt(1) - state.quat[..., 3] ** 2
    cost = dt(5.0) * pos_err + dt(3.0) * quat_err
    cost = cost * dt(0.02)
    return -cost


REWARD_FNS = {"penyaw": tracking_penyaw_reward_fn, "realworld": tracking_realworld_reward_fn}


# ----------------------------------------------------------------------------
# disturbance models: dynamics/free.py:9-72.  Each returns the f_disturb the NEXT step uses, from the PRE-step state
# (free.py:147).  Randomness is explicit: `draw` stands for the value jax.random would return for this call's key --
#   periodic / mixed: uniform(disturb_key, (3,), -disturb_scale, disturb_scale)      (free.py:16-21)
#   gaussian:         normal(disturb_key, (3,))                                      (free.py:66-70)
# ----------------------------------------------------------------------------
def period_disturb(draw, p: Params, s: State):
    """free.py:10-24."""
    return np.where(s.time % p.disturb_period == 0, np.asarray(draw, dtype=s.pos.dtype), s.f_disturb)


def sin_disturb(p: Params, s: State):
    """free.py:27-38."""
    t = s.pos.dtype.type
    dp = _c(p.disturb_params, s.pos.dtype)
    scale = dp[:3] * t(p.disturb_scale)
    period = dp[:3] * t(p.disturb_period / 3) + t(p.disturb_period)
    phase = dp[3:6] * t(2) * t(np.pi)
    return scale * np.sin(t(2) * t(np.pi) / period * t(s.time) + phase)


def drag_disturb(p: Params, s: State):
    """free.py:41-47."""
    t = s.pos.dtype.type
    rel_vel = s.vel - _c(p.disturb_params, s.pos.dtype)[:3] * t(0.5)
    return -np.abs(t(p.disturb_scale)) * rel_vel * np.abs(rel_vel) / t(1.5 ** 2)


def mixed_disturb(draw, p: Params, s: State):
    """free.py:50-56."""
    t = s.pos.dtype.type
    return (drag_disturb(p, s) + sin_disturb(p, s) + period_disturb(draw, p, s)) / t(3)


DISTURB_KINDS = ("none", "gaussian", "periodic", "sin", "drag", "mixed")


def disturb_func(kind: str, p: Params, s: State, draw=None, deterministic: bool = False):
    """free.py:58-72 + the deterministic switch of envs/quadrotor.py:234-235 (it zeroes dyn_noise_scale ONLY: the
    other models keep acting in deterministic rollouts)."""
    t = s.pos.dtype.type
    if kind == "none":
        return np.zeros(3, dtype=s.pos.dtype)
    if kind == "gaussian":
        scale = t(p.dyn_noise_scale) * t(0.0 if deterministic else 1.0)
        return scale * np.asarray(draw if draw is not None else np.zeros(3), dtype=s.pos.dtype)
    if kind == "periodic":
        return period_disturb(draw, p, s)
    if kind == "sin":
        return sin_disturb(p, s)
    if kind == "drag":
        return drag_disturb(p, s)
    if kind == "mixed":
        return mixed_disturb(draw, p, s)
    raise NotImplementedError(kind)


@dataclass
class Disturb:
    """How step_env gets its next disturbance: the model name, the explicit random input of this call (see above)
    and step_env's `deterministic` argument."""
    kind: str = "none"
    draw: Optional[np.ndarray] = None
    deterministic: bool = False


def is_terminal(state: State, p: Params, rollover: bool = False):
    """envs/quadrotor.py:479-490; rollover = not disable_rollover_terminate (main() passes True, :779 -> rollover False)."""
    done = (state.time >= p.max_steps_in_episode) | np.any(np.abs(state.pos) > 3.0, axis=-1)
    if rollover:  # :486-490
        done = done | (state.quat[..., 3] < np.cos(np.pi / 4.0)) | np.any(np.abs(state.omega) > 100.0, axis=-1)
    return done


# ----------------------------------------------------------------------------
# dynamics: dynamics/free.py
# ----------------------------------------------------------------------------
def quad_dynamics_bodyrate(x, u, p: Params, dt):
    """free.py:74-112.  x = [r(3) q(4) v(3) omega(3) f(3)], u = [thrust, omega_tar(3)]."""
    t = x.dtype.type
    u = u * t(p.action_scale)  # :82
    thrust = u[..., 0]
    omega_tar = u[..., 1:4]
    r = x[..., 0:3]
    q = x[..., 3:7] / norm(x[..., 3:7])[..., None]  # :88
    v = x[..., 7:10]
    omega = x[..., 10:13]
    f = x[..., 13:16]
    Q = qtoQ(q)  # :92
    H = geom_H(x.dtype)
    r_dot = v
    q_dot = t(0.5) * np.einsum("...ij,...j->...i", geom_L(q) @ H, omega)  # :96
    e3T = np.zeros(x.shape[:-1] + (3,), dtype=x.dtype)
    e3T[..., 2] = thrust
    grav = np.asarray([0, 0, -p.g], dtype=x.dtype)
    v_dot = grav + t(1.0) / t(p.m) * (np.einsum("...ij,...j->...i", Q, e3T) + f)  # :97-99
    dt = t(dt)
    r_new = r + r_dot * dt
    q_new = q + q_dot * dt
    v_new = v + v_dot * dt
    a = t(p.alpha_bodyrate)
    omega_new = a * omega + (t(1) - a) * omega_tar  # :105-107
    return np.concatenate([r_new, q_new, v_new, omega_new, f], axis=-1)


def free_dynamics_3d_bodyrate(p: Params, s: State, thrust, torque, f_disturb_next, dt):
    """free.py:114-202.  ``f_disturb_next`` = disturb_func(...) of :147 (explicit)."""
    t = s.pos.dtype.type
    max_torque = _c(p.max_torque, s.pos.dtype)
    max_omega = _c(p.max_omega, s.pos.dtype)
    omega_tar = torque / max_torque * max_omega  # :122
    u = np.concatenate([thrust[..., None], omega_tar], axis=-1)
    bshape = u.shape[:-1]
    bc = lambda a: np.broadcast_to(a, bshape + a.shape[-1:])
    x = np.concatenate([bc(s.pos), bc(s.quat), bc(s.vel), bc(s.omega), bc(s.f_disturb)], -1)
    x_new = quad_dynamics_bodyrate(x, u, p, dt)
    pos = x_new[..., 0:3]
    quat = x_new[..., 3:7] / norm(x_new[..., 3:7])[..., None]  # :139
    vel = x_new[..., 7:10]
    omega = x_new[..., 10:13]
    time = s.time + 1  # :150
    T = s.pos_traj.shape[0]
    idx = min(max(time, 0), T - 1)  # JAX gather clamps out-of-range indices
    return s.replace(
        pos=pos, vel=vel, quat=quat, omega=omega,
        pos_tar=s.pos_traj[idx], vel_tar=s.vel_traj[idx], acc_tar=s.acc_traj[idx],
        time=time, f_disturb=np.asarray(f_disturb_next, dtype=s.pos.dtype),
    )


def raw_step(s: State, sub_action, p: Params, f_disturb_next):
    """envs/quadrotor.py:250-263."""
    t = s.pos.dtype.type
    sub_action = np.clip(sub_action, t(-1.0), t(1.0))
    thrust = (sub_action[..., 0] + t(1.0)) / t(2.0) * t(p.max_thrust)
    torque = sub_action[..., 1:] * _c(p.max_torque, s.pos.dtype)
    return free_dynamics_3d_bodyrate(p, s, thrust, torque, f_disturb_next, p.dt)


def step_env(s: State, action, p: Params, f_disturb_next, rollover: bool = False, reward_fn=tracking_penyaw_reward_fn):
    """envs/quadrotor.py:215-248 (lower_controller='base', substeps=1).

    Reward and termination are evaluated on the PRE-step state (:243-244).
    ``f_disturb_next``: either the explicit next disturbance vector (what disturb_func returned; deterministic
    rollouts of 'none' / 'gaussian': 0) or a `Disturb` (model evaluated on the pre-step state, free.py:147).
    Returns (next_state, reward, done).
    """
    t = s.pos.dtype.type
    action = np.clip(action, t(-1.0), t(1.0))
    if isinstance(f_disturb_next, Disturb):
        f_disturb_next = disturb_func(f_disturb_next.kind, p, s, f_disturb_next.draw, f_disturb_next.deterministic)
    nxt = raw_step(s, action, p, f_disturb_next)
    return nxt, reward_fn(s), is_terminal(s, p, rollover)


# ----------------------------------------------------------------------------
# MPC pieces: controllers/covo.py, controllers/mppi.py
# ----------------------------------------------------------------------------
def shift_mean(a_mean):
    """covo.py:201-203 / mppi.py:43-49."""
    return np.concatenate([a_mean[1:], a_mean[-1:]], axis=0)


def sample_actions_full(a_mean, a_cov, eps):
    """covo.py:212-224: mean + chol_lower(cov) @ eps, reshape (N,H,du), clip."""
    H, du = a_mean.shape
    L = np.linalg.cholesky(a_cov.astype(np.float64)).astype(a_mean.dtype)
    a = a_mean.reshape(-1)[None, :] + eps @ L.T
    return np.clip(a.reshape(eps.shape[0], H, du), -1.0, 1.0).astype(a_mean.dtype), L


def sample_actions_blockdiag(a_mean, a_cov, eps):
    """mppi.py:53-66: per-step 4x4 covariance; eps (N,H,du)."""
    H, du = a_mean.shape
    Ls = np.stack([np.linalg.cholesky(a_cov[t].astype(np.float64)) for t in range(H)]).astype(a_mean.dtype)
    a = a_mean[None] + np.einsum("tij,ntj->nti", Ls, eps)
    return np.clip(a, -1.0, 1.0).astype(a_mean.dtype), Ls


def rollout(s0: State, p: Params, a_sampled, discount, f_disturb_shared, rollover: bool = False,
            reward_fn=tracking_penyaw_reward_fn):
    """covo.py:227-263 / mppi.py:71-106.

    a_sampled (N,H,du) already clipped.  ``f_disturb_shared``: (3,) the single
    vector every sample/step receives from the shared ``step_key`` (0 for CoVO's
    deterministic=True and for disturb_type='none'; 0.05*z for MPPI+gaussian), or a
    `Disturb` whose ``draw`` is that one shared draw (every step_env call of the scan
    gets the SAME key, covo.py:225,231): periodic / sin / drag / mixed.
    Returns cost (N,), rewards (N,H), poses (H,N,3).
    """
    dtype = s0.pos.dtype
    N, H, _ = a_sampled.shape
    s = s0
    reward_before = np.zeros(N, dtype=dtype)
    done_before = np.zeros(N, dtype=bool)
    rewards = np.zeros((N, H), dtype=dtype)
    poses = np.zeros((H, N, 3), dtype=dtype)
    for k in range(H):
        s, reward, done = step_env(s, a_sampled[:, k], p, f_disturb_shared, rollover, reward_fn)
        reward = np.where(done_before, reward_before, reward)  # covo.py:233
        done_before = done | done_before
        reward_before = reward
        rewards[:, k] = reward
        poses[k] = s.pos
    disc = np.power(dtype.type(discount), np.arange(H)).astype(dtype)
    cost = -np.sum(rewards * disc, axis=1)  # covo.py:256-263
    return cost.astype(dtype), rewards, poses


def softmax_update(cost, a_sampled, lam, gamma_mean, a_mean):
    """covo.py:266-278."""
    dtype = a_sampled.dtype
    cost_exp = np.exp(-(cost - np.min(cost)) / dtype.type(lam))
    weight = cost_exp / np.sum(cost_exp)
    a_new = np.sum(weight[:, None, None] * a_sampled, axis=0) * dtype.type(gamma_mean) + a_mean * (
        dtype.type(1) - dtype.type(gamma_mean)
    )
    return a_new.astype(dtype), weight


def mppi_cov_update(weight, a_sampled, a_mean_new, a_cov, gamma_sigma):
    """mppi.py:119-125 (uses the NEW mean)."""
    d = a_sampled - a_mean_new[None]
    outer = d[..., :, None] * d[..., None, :]
    return np.sum(weight[:, None, None, None] * outer, axis=0) * gamma_sigma + a_cov * (1 - gamma_sigma)


def softmax_partial(cost, a_flat, lam):
    """Online-softmax partial (m, s, v) of one shard."""
    m = np.min(cost)
    w = np.exp(-(cost - m) / cost.dtype.type(lam))
    return m, np.sum(w), w @ a_flat


def merge_partials(ms, ss, vs, lam, gamma_mean, a_mean):
    """Merge shard partials; mathematically identical to covo.py:266-275."""
    ms = np.asarray(ms); ss = np.asarray(ss); vs = np.asarray(vs)
    m = np.min(ms)
    scale = np.exp(-(ms - m) / ms.dtype.type(lam))
    s = np.sum(ss * scale)
    v = np.sum(vs * scale[:, None], axis=0)
    return (v / s).reshape(a_mean.shape) * gamma_mean + a_mean * (1 - gamma_mean)


def softmax_partial_cov(cost, a_sampled, lam, mu):
    """The same with MPPI's second moments about `mu` (the shifted OLD mean every shard knows; SURVEY 5.8 + mppi.py:119-125):
    S2[t] = sum_n w_n d_n[t] d_n[t]^T, d = a - mu.  a_sampled (n, H, du) -> (m, s, v (H*du,), S2 (H, du, du))."""
    m = np.min(cost)
    w = np.exp(-(cost - m) / cost.dtype.type(lam))
    d = a_sampled - mu[None]
    S2 = np.einsum("n,nti,ntj->tij", w, d, d)
    return m, np.sum(w), w @ a_sampled.reshape(len(cost), -1), S2


def merge_partials_cov(ms, ss, vs, S2s, lam, gamma_mean, a_mean, a_cov, gamma_sigma):
    """Merge the shards' partials with second moments -> (new mean, adapted covariances); mathematically identical to
    mppi.py:109-125 on the unsharded samples: sum_n w_n (a_n - mean')(a_n - mean')^T with the NEW mean expands, with d = a - mu,
    e = mean' - mu, m1 = sum_n w_n d_n and sum_n w_n = 1, to  S2 - m1 e^T - e m1^T + e e^T."""
    ms = np.asarray(ms); ss = np.asarray(ss); vs = np.asarray(vs); S2s = np.asarray(S2s)
    m = np.min(ms)
    scale = np.exp(-(ms - m) / ms.dtype.type(lam))
    s = np.sum(ss * scale)
    wmean = (np.sum(vs * scale[:, None], axis=0) / s).reshape(a_mean.shape)
    S2 = np.sum(S2s * scale[:, None, None, None], axis=0) / s
    mean_new = wmean * gamma_mean + a_mean * (1 - gamma_mean)
    m1, e = wmean - a_mean, mean_new - a_mean
    C = S2 - m1[:, :, None] * e[:, None, :] - e[:, :, None] * m1[:, None, :] + e[:, :, None] * e[:, None, :]
    return mean_new, C * gamma_sigma + a_cov * (1 - gamma_sigma)


def pos_stats(poses):
    """covo.py:281: mean / population std over samples of post-step positions."""
    return np.mean(poses, axis=1), np.std(poses, axis=1)


def optimize_sigma(R, sample_sigma, H, du):
    """covo.py:116-132."""
    R = (R + R.T) / 2.0
    eigns, u = np.linalg.eigh(R)
    offset = -np.min(eigns) + 1e-2
    eigns = eigns + offset
    log_o = np.log(eigns)
    n = du * H
    log_det_a_cov = n * (np.log(sample_sigma) * 2)
    log_const = (log_det_a_cov * 2 + np.sum(log_o)) / n
    log_s = 0.5 * log_const - 0.5 * log_o
    a_cov = u @ np.diag(np.exp(log_s)) @ u.T
    return (a_cov + a_cov.T) / 2.0


def hessian_objective(s0: State, p: Params, a_flat, H, reward_fn=tracking_penyaw_reward_fn, kind: str = "none", draws=None):
    """covo.py:165-180: -(sum_k r(s_k) + r(s_0)); deterministic=True, no discount/freeze.  ``draws`` (H,3): the
    per-step random inputs of the disturbance model (get_hessian splits its key once per step, covo.py:151)."""
    a = a_flat.reshape(H, -1)
    s = s0
    total = s0.pos.dtype.type(0)
    for i in range(H):
        d = Disturb(kind, None if draws is None else draws[i], deterministic=True)
        s, reward, _ = step_env(s, a[i], p, d, reward_fn=reward_fn)
        total = total + reward
    total = total + reward_fn(s0)
    return -total


def hessian_fd(s0: State, p: Params, a_flat, H, h=1e-4, **kw):
    """Central finite-difference Hessian of ``hessian_objective`` (fp64 only;
    O(h^2) truncation -- a coarse cross-check of the AD oracle)."""
    n = a_flat.size
    f = lambda a: hessian_objective(s0, p, a, H, **kw)
    R = np.zeros((n, n))
    f0 = f(a_flat)
    E = np.eye(n) * h
    fp = np.array([f(a_flat + E[i]) for i in range(n)])
    fm = np.array([f(a_flat - E[i]) for i in range(n)])
    for i in range(n):
        R[i, i] = (fp[i] - 2 * f0 + fm[i]) / h**2
        for j in range(i + 1, n):
            fpp = f(a_flat + E[i] + E[j])
            fmm = f(a_flat - E[i] - E[j])
            R[i, j] = R[j, i] = (fpp - fp[i] - fp[j] + 2 * f0 - fm[i] - fm[j] + fmm) / (2 * h**2)
    return R


def covo_call(s_noisy: State, p: Params, a_mean, a_cov, eps, lam, gamma_mean, discount):
    """covo.py:187-283 with Sigma given (offline lookup or precomputed online Sigma).

    Returns u, a_mean', dict(cost, a_sampled, weight, L, pos_mean, pos_std).
    ``a_mean`` must already be shifted (covo.py:201-203) by the caller.
    """
    a_sampled, L = sample_actions_full(a_mean, a_cov, eps)
    zero = np.zeros(3, dtype=a_mean.dtype)
    cost, rewards, poses = rollout(s_noisy, p, a_sampled, discount, zero)
    a_new, weight = softmax_update(cost, a_sampled, lam, gamma_mean, a_mean)
    pm, ps = pos_stats(poses)
    return a_new[0], a_new, dict(cost=cost, a_sampled=a_sampled, weight=weight, L=L,
                                 pos_mean=pm, pos_std=ps, rewards=rewards)


def mppi_call(s_noisy: State, p: Params, a_mean, a_cov, eps, lam, gamma_mean, gamma_sigma,
              discount, f_disturb_shared):
    """mppi.py:28-134; ``a_mean``/``a_cov`` already shifted (mppi.py:43-49)."""
    a_sampled, Ls = sample_actions_blockdiag(a_mean, a_cov, eps)
    cost, rewards, poses = rollout(s_noisy, p, a_sampled, discount, f_disturb_shared)
    a_new, weight = softmax_update(cost, a_sampled, lam, gamma_mean, a_mean)
    cov_new = mppi_cov_update(weight, a_sampled, a_new, a_cov, gamma_sigma)
    pm, ps = pos_stats(poses)
    return a_new[0], a_new, cov_new, dict(cost=cost, a_sampled=a_sampled, weight=weight,
                                           pos_mean=pm, pos_std=ps)


# ----------------------------------------------------------------------------
# environment plumbing: envs/quadrotor.py
# ----------------------------------------------------------------------------
def hover_action(p: Params, H, dtype=np.float32):
    """envs/quadrotor.py:685-690."""
    th = (p.m * p.g / p.max_thrust) * 2.0 - 1.0
    return np.tile(np.asarray([th, 0.0, 0.0, 0.0], dtype=dtype), (H, 1))


def zero_state(pos_traj, vel_traj, acc_traj, f_disturb, dtype=np.float32) -> State:
    """envs/quadrotor.py:265-312."""
    z = np.zeros(3, dtype=dtype)
    return State(
        pos=z.copy(), vel=z.copy(), quat=np.asarray([0, 0, 0, 1], dtype=dtype), omega=z.copy(),
        f_disturb=np.asarray(f_disturb, dtype=dtype),
        pos_tar=np.asarray(pos_traj[0], dtype=dtype), vel_tar=np.asarray(vel_traj[0], dtype=dtype),
        acc_tar=np.asarray(acc_traj[0], dtype=dtype), time=0,
        pos_traj=np.asarray(pos_traj, dtype=dtype), vel_traj=np.asarray(vel_traj, dtype=dtype),
        acc_traj=np.asarray(acc_traj, dtype=dtype),
    )


def noisy_state(next_state: State, p: Params, z_pos, z_vel, z_quat, z_omega) -> State:
    """envs/quadrotor.py:322-350 (z_* are the standard normals jax would draw)."""
    t = next_state.pos.dtype.type
    sc = t(p.obs_noise_scale)
    return next_state.replace(
        pos=next_state.pos + z_pos * sc * t(0.25),
        vel=next_state.vel + z_vel * sc * t(0.5),
        quat=next_state.quat + z_quat * sc * t(0.02),
        omega=next_state.omega + z_omega * sc * t(0.5),
    )


def generate_fixed_traj(max_steps, dt, rng=None):
    """utils.py:49-53."""
    z = np.zeros((max_steps, 3))
    return z, z.copy(), z.copy()


def generate_lissa_traj(max_steps, dt, rng: np.random.Generator):
    """utils.py:87-130."""
    rand_amp = rng.uniform(-1.0, 1.0, size=(3, 2))
    rand_phase = rng.uniform(-np.pi, np.pi, size=(3, 2))
    ts = np.arange(0, max_steps + 50) * dt
    w1 = 2 * np.pi * 0.2
    w2 = 2 * np.pi * 0.4
    pos = np.stack([rand_amp[i, 0] * np.sin(w1 * ts + rand_phase[i, 0])
                    + rand_amp[i, 1] * np.sin(w2 * ts + rand_phase[i, 1]) for i in range(3)], axis=1)
    pos = pos - pos[0]
    vel = np.stack([rand_amp[i, 0] * w1 * np.cos(w1 * ts + rand_phase[i, 0])
                    + rand_amp[i, 1] * w2 * np.cos(w2 * ts + rand_phase[i, 1]) for i in range(3)], axis=1)
    acc = np.stack([-rand_amp[i, 0] * w1**2 * np.sin(w1 * ts + rand_phase[i, 0])
                    - rand_amp[i, 1] * w2**2 * np.sin(w2 * ts + rand_phase[i, 1]) for i in range(3)], axis=1)
    return pos, vel, acc


def generate_zigzag_traj(max_steps, dt, rng: np.random.Generator):
    """utils.py:183-251.  Key reuse quirks of the reference (identical key arrays
    for key-points and angles, segments 0 and 1 sharing keys[1]) cannot be
    reproduced bit-wise without jax's PRNG; the draw *structure* is kept:
    one (distance, dtheta, dphi) triple per key, segments 0 and 1 share a triple."""
    point_per_seg = 40
    num_seg = max_steps // point_per_seg + 1
    prev_point = rng.uniform(-1.0, 1.0, size=3)
    prev_point = prev_point / np.linalg.norm(prev_point) * 0.1
    draws = [(rng.uniform(1.0, 1.5), rng.uniform(-np.pi / 3, np.pi / 3, size=2)) for _ in range(num_seg + 1)]
    pos_segs, vel_segs = [], []
    for i in range(num_seg):
        # carry starts with keys[1] (:241) and segment i hands keys[i+1] to segment i+1 (:238)
        distance, (dth, dph) = draws[1] if i == 0 else draws[i]
        vec_to_center = -prev_point / np.linalg.norm(prev_point)
        theta = np.arccos(vec_to_center[2]) + dth
        phi = np.arctan2(vec_to_center[1], vec_to_center[0]) + dph
        new_dir = np.array([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)])
        next_point = prev_point + distance * new_dir
        seg = np.stack([np.linspace(a, b, point_per_seg, endpoint=False)
                        for a, b in zip(prev_point, next_point)], axis=-1)
        dseg = (next_point - prev_point) / (point_per_seg + 1) * np.ones((point_per_seg, 3)) / dt
        pos_segs.append(seg)
        vel_segs.append(dseg)
        prev_point = next_point
    pos = np.concatenate(pos_segs, axis=0)
    pos = pos - pos[0]
    vel = np.concatenate(vel_segs, axis=0)
    return pos, vel, np.zeros_like(pos)


def pid_action(s: State, p_default: Params, Kp=10.0, Kd=5.0, Ki=0.0, Kp_att=10.0, integral=None):
    """controllers/pid.py:38-84 (uses the DEFAULT params' m, g: pid.py:33,44)."""
    dt = s.pos.dtype
    integral = np.zeros(3, dtype=dt) if integral is None else integral
    Q = qtoQ(s.quat)
    f_d = p_default.m * (np.array([0.0, 0.0, p_default.g], dtype=dt) - Kp * (s.pos - s.pos_tar)
                         - Kd * (s.vel - s.vel_tar) - Ki * integral + s.acc_tar)
    thrust = np.clip((Q.T @ f_d)[2], 0.0, p_default.max_thrust)
    f_d_norm = norm(f_d)
    f_d_norm = np.where(f_d_norm < 1e-3, 1e-3, f_d_norm)
    z_d = f_d / f_d_norm
    axis_angle = np.cross(np.array([0.0, 0.0, 1.0], dtype=dt), z_d)
    angle = norm(axis_angle)
    angle = np.where(angle < 1e-3, 5e-4, angle)
    axis = np.where(angle < 1e-3, np.array([0.0, 0.0, 1.0], dtype=dt), axis_angle / angle)
    R_d = axisangletoR(axis, angle)
    R_e = R_d.T @ Q
    angle_err = vee(R_e - R_e.T)
    omega_d = -Kp_att * angle_err
    action = np.concatenate([[thrust / p_default.max_thrust * 2.0 - 1.0],
                             omega_d / np.asarray(p_default.max_omega, dtype=dt)]).astype(dt)
    return action, integral + (s.pos - s.pos_tar) * p_default.dt
