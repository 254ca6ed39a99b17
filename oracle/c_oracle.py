"""ctypes binding of oracle/_build/libcovo_oracle.so.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/__init__.py).  Builds the library on first use.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcovo_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("covo_oracle.c", "covo_oracle_body.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def params_vec(p) -> np.ndarray:
    """[max_thrust, max_torque(3), max_omega(3), dt, g, m, action_scale, alpha_bodyrate]"""
    return np.asarray([p.max_thrust, *p.max_torque, *p.max_omega, p.dt, p.g, p.m, p.action_scale,
                       p.alpha_bodyrate], dtype=np.float64)


def state22(s, dtype) -> np.ndarray:
    return np.concatenate([s.pos, s.vel, s.quat, s.omega, s.f_disturb, s.pos_tar, s.vel_tar]).astype(dtype)


REWARD_KINDS = {"penyaw": 0, "realworld": 1}
DISTURB_KINDS = {"none": 0, "gaussian": 1, "periodic": 2, "sin": 3, "drag": 4, "mixed": 5}


def dist_vec(p, kind: str) -> np.ndarray:
    """[kind, disturb_period, disturb_scale, disturb_params[6]] (dynamics/dataclass.py:86-88)"""
    return np.asarray([DISTURB_KINDS[kind], p.disturb_period, p.disturb_scale, *p.disturb_params], dtype=np.float64)


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct)) if a is not None else None


def rollout(s, p, a_sampled, discount=1.0, f_shared=None, dtype=np.float32, want_rewards=False, want_poses=False,
            rollover=False, reward="penyaw", disturb=None):
    """a_sampled (N,H,4).  Returns cost[, rewards (N,H)][, poses (H,N,3)].  rollover: is_terminal's rollover test
    (envs/quadrotor.py:486-490, Quad3D(disable_rollover_terminate=False)).  reward: "penyaw" (utils.py:285-294) or
    "realworld" (:297-313).  disturb: None (f_shared is the explicit next disturbance of every step) or a ref_np.Disturb
    with kind periodic / sin / drag / mixed and the ONE shared draw (free.py:10-58)."""
    ct = C.c_float if dtype == np.float32 else C.c_double
    fn = lib().oracle_rollout_ex_f32 if dtype == np.float32 else lib().oracle_rollout_ex_f64
    a = np.ascontiguousarray(a_sampled, dtype=dtype)
    N, H, _ = a.shape
    prm = params_vec(p)
    st = state22(s, dtype)
    pt = np.ascontiguousarray(s.pos_traj, dtype=dtype)
    vt = np.ascontiguousarray(s.vel_traj, dtype=dtype)
    dist = None
    if disturb is not None and disturb.kind not in ("none", "gaussian"):
        dist = dist_vec(p, disturb.kind)
        f_shared = np.zeros(3) if disturb.draw is None else disturb.draw
    elif disturb is not None and disturb.kind == "gaussian" and f_shared is None:
        f_shared = (0.0 if disturb.deterministic else p.dyn_noise_scale) * np.asarray(disturb.draw)
    fs = np.zeros(3, dtype=dtype) if f_shared is None else np.ascontiguousarray(f_shared, dtype=dtype)
    cost = np.empty(N, dtype=dtype)
    rew = np.empty((N, H), dtype=dtype) if want_rewards else None
    pos = np.empty((H, N, 3), dtype=dtype) if want_poses else None
    fn.restype = None
    fn(_p(prm, C.c_double), C.c_int(p.max_steps_in_episode), _p(st, ct), C.c_int(int(s.time)), _p(pt, ct), _p(vt, ct),
       C.c_int(pt.shape[0]), _p(a, ct), C.c_long(N), C.c_int(H), ct(discount), _p(fs, ct), _p(cost, ct), _p(rew, ct),
       _p(pos, ct), C.c_int(1 if rollover else 0), C.c_int(REWARD_KINDS[reward]), _p(dist, C.c_double))
    out = [cost]
    if want_rewards:
        out.append(rew)
    if want_poses:
        out.append(pos)
    return out[0] if len(out) == 1 else tuple(out)


def noise_gemm(L, mu, eps):
    """a = clip(mu + L @ eps) as an ascending-k fmaf chain.  eps (N,n) -> a (N,n) fp32."""
    L = np.ascontiguousarray(L, dtype=np.float32)
    mu = np.ascontiguousarray(mu, dtype=np.float32).reshape(-1)
    eps = np.ascontiguousarray(eps, dtype=np.float32)
    N, n = eps.shape
    a = np.empty((N, n), dtype=np.float32)
    f = lib().oracle_noise_gemm_f32
    f.restype = None
    f(_p(L, C.c_float), _p(mu, C.c_float), _p(eps, C.c_float), C.c_long(N), C.c_int(n), _p(a, C.c_float))
    return a


def noise_blockdiag(Ls, mu, eps):
    Ls = np.ascontiguousarray(Ls, dtype=np.float32)
    mu = np.ascontiguousarray(mu, dtype=np.float32)
    eps = np.ascontiguousarray(eps, dtype=np.float32)
    N, H, _ = eps.shape
    a = np.empty((N, H, 4), dtype=np.float32)
    f = lib().oracle_noise_blockdiag_f32
    f.restype = None
    f(_p(Ls, C.c_float), _p(mu, C.c_float), _p(eps, C.c_float), C.c_long(N), C.c_int(H), _p(a, C.c_float))
    return a


def softmax_partial(cost, a_flat, lam, dtype=np.float32):
    ct = C.c_float if dtype == np.float32 else C.c_double
    fn = lib().oracle_softmax_partial_f32 if dtype == np.float32 else lib().oracle_softmax_partial_f64
    cost = np.ascontiguousarray(cost, dtype=dtype)
    a = np.ascontiguousarray(a_flat, dtype=dtype)
    N, n = a.shape
    m = ct()
    s = ct()
    v = np.empty(n, dtype=dtype)
    fn.restype = None
    fn(_p(cost, ct), _p(a, ct), C.c_long(N), C.c_int(n), ct(lam), C.byref(m), C.byref(s), _p(v, ct))
    return dtype(m.value), dtype(s.value), v


def sampling_step(s, p, L, a_mean, eps, lam, gamma_mean=1.0, discount=1.0, threads=None):
    """Whole fp32 sampling step (noise GEMM -> rollout -> softmax update); the timed CPU baseline."""
    if threads is not None:
        os.environ["OMP_NUM_THREADS"] = str(threads)
    eps = np.ascontiguousarray(eps, dtype=np.float32)
    N, n = eps.shape
    H = n // 4
    L = np.ascontiguousarray(L, dtype=np.float32)
    am = np.ascontiguousarray(a_mean, dtype=np.float32).reshape(-1)
    prm = params_vec(p)
    st = state22(s, np.float32)
    pt = np.ascontiguousarray(s.pos_traj, dtype=np.float32)
    vt = np.ascontiguousarray(s.vel_traj, dtype=np.float32)
    a_work = np.empty((N, n), dtype=np.float32)
    cost = np.empty(N, dtype=np.float32)
    out = np.empty(n, dtype=np.float32)
    f = lib().oracle_sampling_step_f32
    f.restype = None
    cf = C.c_float
    f(_p(prm, C.c_double), C.c_int(p.max_steps_in_episode), _p(st, cf), C.c_int(int(s.time)), _p(pt, cf), _p(vt, cf),
      C.c_int(pt.shape[0]), _p(L, cf), _p(am, cf), _p(eps, cf), C.c_long(N), C.c_int(H), cf(lam), cf(gamma_mean),
      cf(discount), _p(a_work, cf), _p(cost, cf), _p(out, cf))
    return out.reshape(H, 4), cost, a_work


def hessian(s, p, a_flat, H=32, threads=None, reward="penyaw", kind="none", draws=None, table=None):
    """covo.py:134-185 by hyper-dual forward-over-forward AD in C (fp64, OpenMP over the n(n+1)/2 pairs).
    kind / draws (H,3): the disturbance model and its per-step uniform draws (free.py:10-58; get_hessian splits its key
    once per step, covo.py:151).  table (H,4): the per-step force table given explicitly instead (rows {g_k, c_k} of
    include/covo_hip.h's covo_disturb_table; `kind` then only selects the drag coefficient)."""
    if threads is not None:
        os.environ["OMP_NUM_THREADS"] = str(threads)
    prm = params_vec(p)
    st = state22(s, np.float64)
    tb = np.ascontiguousarray(table, dtype=np.float64).reshape(H, 4) if table is not None else None
    pt = np.ascontiguousarray(s.pos_traj, dtype=np.float64)
    vt = np.ascontiguousarray(s.vel_traj, dtype=np.float64)
    a = np.ascontiguousarray(a_flat, dtype=np.float64).reshape(-1)
    n = 4 * H
    R = np.zeros((n, n), dtype=np.float64)
    dist = dist_vec(p, kind) if kind not in ("none", "gaussian") else None
    dr = np.ascontiguousarray(draws, dtype=np.float64).reshape(H, 3) if draws is not None else None
    f = lib().oracle_hessian_ex_f64
    f.restype = None
    cd = C.c_double
    f(_p(prm, cd), _p(st, cd), C.c_int(int(s.time)), _p(pt, cd), _p(vt, cd), C.c_int(pt.shape[0]), _p(a, cd), C.c_int(H),
      _p(R, cd), C.c_int(REWARD_KINDS[reward]), _p(dist, cd), _p(dr, cd), _p(tb, cd))
    return R
