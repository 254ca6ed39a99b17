"""torch fp64 restatement of CoVO's Hessian objective + exact AD Hessian.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/__init__.py).  Stands in for
    jax.jacfwd(jax.jacfwd(get_cumulated_cost))      (quadjax/controllers/covo.py:165-185)
using torch.func forward-over-forward AD on a differentiable restatement of
envs/quadrotor.py:215-263, dynamics/free.py:74-155, dynamics/utils.py:266-294.
torch's JVP conventions match jax's for the primitives used (abs' = sign, norm'(0) = nan) except an
exact clip/min/max tie, which jax splits 0.5/0.5; `tie_half=True` reproduces that.
"""
import numpy as np
import torch


def _clip(x, lo, hi, tie_half=True):
    if not tie_half:
        return torch.clamp(x, lo, hi)
    # jnp.clip = minimum(maximum(x, lo), hi); lax.max/min JVP at a tie = 0.5
    glo = torch.where(x > lo, 1.0, torch.where(x == lo, 0.5, 0.0)).detach()
    y = lo + (x - lo) * glo
    y = torch.where(x < lo, torch.full_like(x, lo), y)
    ghi = torch.where(y < hi, 1.0, torch.where(y == hi, 0.5, 0.0)).detach()
    z = hi + (y - hi) * ghi
    return torch.where(y > hi, torch.full_like(y, hi), z)


def reward(pos, vel, quat, pos_tar, vel_tar):
    """utils.py:285-294."""
    err_pos = torch.linalg.norm(pos_tar - pos)
    err_vel = torch.linalg.norm(vel_tar - vel)
    yaw = torch.atan2(2 * (quat[3] * quat[2] + quat[0] * quat[1]), 1 - 2 * (quat[1] ** 2 + quat[2] ** 2))
    l = torch.log(err_pos + 1)
    lp = (err_pos * 0.4 + _clip(l * 4, 0.0, 1.0) * 0.4 + _clip(l * 8, 0.0, 1.0) * 0.2 + _clip(l * 16, 0.0, 1.0) * 0.1
          + _clip(l * 32, 0.0, 1.0) * 0.1)
    return 1.3 - 0.05 * err_vel - lp - torch.abs(yaw) * 0.2


def reward_realworld(pos, quat, pos_tar):
    """utils.py:297-313."""
    pos_err = torch.mean((pos - pos_tar) ** 2)
    quat_err = 1 - quat[3] ** 2
    return -((5.0 * pos_err + 3.0 * quat_err) * 0.02)


def disturb_next(kind, prm, time, vel, f, draw):
    """free.py:10-58 on the PRE-step state (deterministic=True: 'gaussian' is 0, quadrotor.py:234)."""
    dt = vel.dtype
    dp = prm["disturb_params"]
    drag = -abs(prm["disturb_scale"]) * (vel - dp[:3] * 0.5) * torch.abs(vel - dp[:3] * 0.5) / (1.5 ** 2)
    period = dp[:3] * (prm["disturb_period"] / 3) + prm["disturb_period"]
    sn = dp[:3] * prm["disturb_scale"] * torch.sin(2 * torch.pi / period * time + dp[3:6] * 2 * torch.pi)
    if kind not in ("periodic", "sin", "drag", "mixed"):
        return torch.zeros(3, dtype=dt)
    per = torch.as_tensor(np.zeros(3) if draw is None else draw, dtype=dt) if time % prm["disturb_period"] == 0 else f
    if kind == "periodic":
        return per
    if kind == "sin":
        return sn
    if kind == "drag":
        return drag
    if kind == "mixed":
        return (drag + sn + per) / 3
    return torch.zeros(3, dtype=dt)


def dyn(pos, vel, quat, omega, f, act, prm):
    """quadrotor.py:250-263 + free.py:74-139 (closed-form geometry)."""
    act = _clip(_clip(act, -1.0, 1.0), -1.0, 1.0)  # quadrotor.py:223 and :258
    thrust = (act[0] + 1.0) / 2.0 * prm["max_thrust"] * prm["action_scale"]
    omega_tar = act[1:] * prm["max_torque"] / prm["max_torque"] * prm["max_omega"] * prm["action_scale"]
    q = quat / torch.linalg.norm(quat)
    x, y, z, w = q[0], q[1], q[2], q[3]
    Qz = torch.stack([2 * (x * z + y * w), 2 * (y * z - x * w), w * w - x * x - y * y + z * z])
    v3 = q[:3]
    qd = 0.5 * torch.cat([w * omega + torch.linalg.cross(v3, omega), -(v3 * omega).sum().reshape(1)])
    g = torch.tensor([0.0, 0.0, -prm["g"]], dtype=pos.dtype)
    vd = g + 1.0 / prm["m"] * (Qz * thrust + f)
    dt = prm["dt"]
    pos_n = pos + vel * dt
    q_n = q + qd * dt
    vel_n = vel + vd * dt
    a = prm["alpha_bodyrate"]
    omega_n = a * omega + (1 - a) * omega_tar
    return pos_n, vel_n, q_n / torch.linalg.norm(q_n), omega_n


def make_objective(s, p, H, reward_kind="penyaw", kind="none", draws=None):
    """Returns f(a_flat) = -(sum_k r(s_k) + r(s_0)) for oracle state `s` (ref_np.State) and params `p`."""
    dt = torch.float64
    t = lambda x: torch.as_tensor(x, dtype=dt)
    prm = dict(max_thrust=float(p.max_thrust), max_torque=t(p.max_torque), max_omega=t(p.max_omega), dt=float(p.dt),
               g=float(p.g), m=float(p.m), action_scale=float(p.action_scale), alpha_bodyrate=float(p.alpha_bodyrate),
               disturb_period=int(p.disturb_period), disturb_scale=float(p.disturb_scale), disturb_params=t(p.disturb_params))
    pos0, vel0, quat0, om0, f0 = t(s.pos), t(s.vel), t(s.quat), t(s.omega), t(s.f_disturb)
    pt, vt = t(s.pos_traj), t(s.vel_traj)
    T = pt.shape[0]
    tar0 = (t(s.pos_tar), t(s.vel_tar))
    rew = (lambda pos, vel, quat, tar: reward(pos, vel, quat, tar[0], tar[1])) if reward_kind == "penyaw" else (
        lambda pos, vel, quat, tar: reward_realworld(pos, quat, tar[0]))

    def f(a_flat):
        a = a_flat.reshape(H, -1)
        pos, vel, quat, om, fd = pos0, vel0, quat0, om0, f0
        total = torch.zeros((), dtype=dt)
        tar = tar0
        for k in range(H):
            total = total + rew(pos, vel, quat, tar)
            fd_next = disturb_next(kind, prm, int(s.time) + k, vel, fd, None if draws is None else draws[k])  # free.py:147
            pos, vel, quat, om = dyn(pos, vel, quat, om, fd, a[k], prm)
            fd = fd_next
            idx = min(max(int(s.time) + k + 1, 0), T - 1)
            tar = (pt[idx], vt[idx])
        total = total + rew(pos0, vel0, quat0, tar0)  # covo.py:176-178
        return -total

    return f


def hessian(s, p, a_flat, H, reward_kind="penyaw", kind="none", draws=None):
    """Exact Hessian (n,n) fp64 by forward-over-forward AD."""
    f = make_objective(s, p, H, reward_kind, kind, draws)
    a = torch.as_tensor(a_flat, dtype=torch.float64).reshape(-1)
    return torch.func.jacfwd(torch.func.jacfwd(f))(a).numpy()
