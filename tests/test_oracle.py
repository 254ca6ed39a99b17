"""CPU tests of the oracle (test infrastructure) against the hand-derived known answers and the
committed golden vectors.  PARITY UNPINNED: no reference-produced fixture exists."""
import json
import os

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import ref_np as R
from oracle import rng_np
from tests.conftest import make_problem, table_ref

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))


def _rest_state(dtype):
    z = np.zeros((320, 3))
    return R.zero_state(z, z, z, [0, 0, 0], dtype=dtype), R.Params()


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 2e-6)])
def test_kat_hover_freefall_thrust_bodyrate_reward(dtype, tol):
    s, p = _rest_state(dtype)
    a = R.hover_action(p, 32, dtype)
    assert np.allclose(a[0], KAT["hover_action"], atol=1e-6)
    cost, rew, poses = R.rollout(s, p, a[None], 1.0, np.zeros(3, dtype=dtype))
    assert abs(cost[0] - KAT["kat1_hover_cost"]) < 50 * tol * 41.6
    assert np.allclose(rew, KAT["kat1_reward"], atol=10 * tol)
    z3 = np.zeros(3, dtype=dtype)
    down = np.array([-1, 0, 0, 0], dtype=dtype)
    s1, _, _ = R.step_env(s, down, p, z3)
    s2, _, _ = R.step_env(s1, down, p, z3)
    k = KAT["kat2_freefall"]
    assert abs(s1.vel[2] - k["v_z_1"]) < tol and abs(s1.pos[2] - k["p_z_1"]) < tol
    assert abs(s2.vel[2] - k["v_z_2"]) < tol and abs(s2.pos[2] - k["p_z_2"]) < tol
    s1, _, _ = R.step_env(s, np.array([1, 0, 0, 0], dtype=dtype), p, z3)
    assert abs(s1.vel[2] - KAT["kat3_full_thrust_v_z"]) < tol
    ar = a[0].copy()
    ar[1] = 1
    s1, _, _ = R.step_env(s, ar, p, z3)
    s2, _, _ = R.step_env(s1, ar, p, z3)
    k = KAT["kat4_bodyrate"]
    assert abs(s1.omega[0] - k["omega_x_1"]) < 10 * tol and abs(s2.omega[0] - k["omega_x_2"]) < 10 * tol
    assert np.allclose(s1.quat, k["quat_1"], atol=tol) and np.allclose(s2.quat, k["quat_2"], atol=tol)
    assert abs(R.tracking_penyaw_reward_fn(s.replace(pos_tar=np.array([1, 0, 0], dtype=dtype))) - KAT["kat5_reward_err1"]) < tol
    assert abs(R.log_pos_fn(dtype(1.0)) - KAT["kat5_log_pos_1"]) < tol
    sq = s.replace(quat=np.array([0, 0, np.sin(np.pi / 4), np.cos(np.pi / 4)], dtype=dtype))
    assert abs(R.tracking_penyaw_reward_fn(sq) - KAT["kat5_reward_yaw90"]) < tol


def test_kat_softmax_and_sigma():
    lam = 0.01
    cost = np.array([1.0, 1.0 + lam * np.log(2)])
    a = np.zeros((2, 32, 4))
    a[0] += 1
    a_new, w = R.softmax_update(cost, a, lam, 1.0, np.zeros((32, 4)))
    assert np.allclose(w, KAT["kat6_softmax"], atol=1e-12) and np.allclose(a_new, 2 / 3)
    assert np.abs(R.optimize_sigma(np.eye(128) * 3.0, 0.5, 32, 4) - 0.25 * np.eye(128)).max() < 1e-14
    rng = np.random.default_rng(0)
    A = rng.normal(size=(128, 128))
    Rm = A + A.T
    S = R.optimize_sigma(Rm, 0.5, 32, 4)
    B = Rm + (0.01 - np.linalg.eigvalsh(Rm).min()) * np.eye(128)
    M = S @ S @ B
    assert np.abs(M - np.eye(128) * M[0, 0]).max() < 1e-9 and np.abs(S - S.T).max() == 0
    assert abs(np.linalg.slogdet(S)[1] - KAT["kat7_logdet_sigma"]) < 1e-8


def test_kat_done_freeze_and_horizon_past_episode_end():
    # KAT 9: |pos_x| > 3 at step k: r_k is live, later rewards repeat it
    s, p, rng = make_problem(seed=3, time=10)
    s = s.replace(pos=s.pos + np.array([2.95, 0, 0]), vel=s.vel + np.array([4.0, 0, 0]))
    a = np.tile(R.hover_action(p, 32, np.float64)[None], (2, 1, 1))
    cost, rew, poses = R.rollout(s, p, a, 1.0, np.zeros(3))
    kx = int(np.argmax(np.abs(np.concatenate([s.pos[None, :1], poses[:, 0, :1]]))[:, 0] > 3))
    assert 0 < kx < 31 and np.all(rew[0, kx:] == rew[0, kx]) and rew[0, kx - 1] != rew[0, kx]
    # KAT 10: t0 + k >= 300 -> done; targets clamp at the last row
    s, p, rng = make_problem(seed=4, time=290)
    cost, rew, poses = R.rollout(s, p, a, 1.0, np.zeros(3))
    assert np.all(rew[0, 10:] == rew[0, 10]) and rew[0, 9] != rew[0, 10]
    s2 = s.replace(time=330)  # beyond the 320-row zigzag table
    nxt, _, _ = R.step_env(s2, a[0, 0], p, np.zeros(3))
    assert np.array_equal(nxt.pos_tar, s.pos_traj[-1])


def test_philox_known_answers():
    for v in KAT["philox4x32_10"]:
        out = rng_np.philox4x32_10(*v["ctr"], *v["key"])
        assert [hex(int(x)) for x in out] == v["out"]
    z = rng_np.randn(7, 9, 0, 2048, 128)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    # global-id keyed: a shard's rows equal the corresponding rows of the full draw
    assert np.array_equal(rng_np.randn(7, 9, 1024, 256, 128), z[1024:1280])


def test_c_oracle_matches_literal_numpy_and_golden():
    g = np.load(os.path.join(HERE, "golden", "rollout_small.npz"))
    p = R.Params().fp32()  # the golden vectors start from the fp32-rounded parameters
    for name in ("mid", "late", "start"):
        st = g[f"{name}_state"]
        z = np.zeros((1, 3))
        s = R.State(pos=st[0:3], vel=st[3:6], quat=st[6:10], omega=st[10:13], f_disturb=st[13:16], pos_tar=st[16:19],
                    vel_tar=st[19:22], acc_tar=np.zeros(3), time=int(g[f"{name}_time"]), pos_traj=g[f"{name}_pos_traj"],
                    vel_traj=g[f"{name}_vel_traj"], acc_traj=np.zeros_like(g[f"{name}_pos_traj"]))
        a, fs, disc = g[f"{name}_a"], g[f"{name}_f_shared"], float(g[f"{name}_discount"])
        c64, r64, p64 = CO.rollout(s, p, a, disc, fs, dtype=np.float64, want_rewards=True, want_poses=True)
        assert np.abs(c64 - g[f"{name}_cost"]).max() < 1e-11
        assert np.abs(r64 - g[f"{name}_rewards"]).max() < 1e-12 and np.abs(p64 - g[f"{name}_poses"]).max() < 1e-12
        c32 = CO.rollout(s.astype(np.float32), p, a.astype(np.float32), disc, fs, dtype=np.float32)
        assert np.abs((c32 - c64) / np.maximum(np.abs(c64), 1)).max() < 1e-5
    assert np.any(g["start_rewards"][:, -1] == g["start_rewards"][:, -2])  # the freeze case is exercised


def test_c_noise_gemm_and_softmax_partial():
    rng = np.random.default_rng(1)
    A = rng.normal(size=(128, 128))
    L = np.linalg.cholesky(A @ A.T / 128 + 0.1 * np.eye(128)).astype(np.float32)
    mu = rng.normal(size=128).astype(np.float32) * 0.3
    eps = rng.normal(size=(200, 128)).astype(np.float32)
    a = CO.noise_gemm(L, mu, eps)
    ref = np.clip(mu[None].astype(np.float64) + eps.astype(np.float64) @ L.astype(np.float64).T, -1, 1)
    assert np.abs(a - ref).max() < 2e-5
    cost = rng.normal(size=200).astype(np.float64) * 0.05
    m, s, v = CO.softmax_partial(cost, a.astype(np.float64), 0.01, dtype=np.float64)
    m2, s2, v2 = R.softmax_partial(cost, a.astype(np.float64), 0.01)
    assert m == m2 and abs(s - s2) < 1e-12 * s2 and np.abs(v - v2).max() < 1e-12
    # shard invariance of the record merge
    am = np.zeros((32, 4))
    full = R.merge_partials([m2], [s2], [v2], 0.01, 1.0, am)
    for G in (2, 4, 8):
        recs = [R.softmax_partial(cost[i::G], a[i::G].astype(np.float64), 0.01) for i in range(G)]
        sh = R.merge_partials([r[0] for r in recs], [r[1] for r in recs], [r[2] for r in recs], 0.01, 1.0, am)
        assert np.abs(sh - full).max() < 1e-12
    ref_mean, _ = R.softmax_update(cost, a.reshape(200, 32, 4).astype(np.float64), 0.01, 1.0, am)
    assert np.abs(full - ref_mean).max() < 1e-12


def test_hessian_ad_oracle_structure():
    """KAT 8: symmetric, exact zeros for the last action, agrees with finite differences."""
    from oracle import ref_torch as RT
    s, p, rng = make_problem(seed=0, time=37)
    H = 4  # short horizon keeps this CPU test to seconds; the GPU suite checks H=32
    a = (R.hover_action(p, H, np.float64) + 0.1 * rng.normal(size=(H, 4))).reshape(-1)
    Rm = RT.hessian(s, p, a, H)
    assert np.abs(Rm - Rm.T).max() < 1e-12 and np.abs(Rm[-4:]).max() == 0.0
    Rfd = R.hessian_fd(s, p, a, H, h=1e-4)
    assert np.abs(Rfd - Rm).max() < 1e-5 * max(1.0, np.abs(Rm).max())


def test_c_hessian_matches_torch_ad():
    """oracle/covo_oracle.c::oracle_hessian_f64 (hyper-dual, the CPU-baseline Hessian) against the torch
    forward-over-forward AD oracle, including two exact clip ties (gradient factor 0.25 through the two clips)."""
    from oracle import c_oracle as CO
    from oracle import ref_torch as RT

    s, p, rng = make_problem(seed=0, time=37)
    a = (R.hover_action(p, 32, np.float64) + 0.1 * rng.normal(size=(32, 4))).astype(np.float32)
    a[3, 1] = 1.0
    a[5, 2] = -1.0
    Rc = CO.hessian(s, p, a.reshape(-1).astype(np.float64), 32)
    Rt = RT.hessian(s, p, a.reshape(-1).astype(np.float64), 32)
    assert np.abs(Rc - Rt).max() < 1e-12 and np.abs(Rc - Rc.T).max() == 0.0 and np.abs(Rc[124:]).max() == 0.0
    assert np.abs(Rt[13]).max() > 0


def test_rollover_termination_oracle():
    """envs/quadrotor.py:486-490: with the rollover test on, a state tipped past 90 degrees (quat[3] < cos(pi/4)) or spinning
    faster than 100 rad/s is terminal; numpy and C restatements agree, and the flag changes nothing for upright samples."""
    from oracle import c_oracle as CO
    s, p, rng = make_problem(seed=5, time=20)
    assert not R.is_terminal(s, p, rollover=True)
    tipped = s.replace(quat=np.array([0.8, 0.0, 0.0, 0.6]))
    assert R.is_terminal(tipped, p, rollover=True) and not R.is_terminal(tipped, p, rollover=False)
    spinning = s.replace(omega=np.array([0.0, -100.5, 0.0]))
    assert R.is_terminal(spinning, p, rollover=True) and not R.is_terminal(spinning, p)
    edge = s.replace(quat=np.array([0.0, 0.0, np.sqrt(0.5), np.cos(np.pi / 4)]))
    assert not R.is_terminal(edge, p, rollover=True)  # strict <
    N = 64
    a = np.clip(R.hover_action(p, 32, np.float64)[None] + np.array([0.3, 1.5, 1.5, 0.5]) * rng.normal(size=(N, 32, 4)), -1, 1)
    c_np, rew_np, _ = R.rollout(s, p, a, 0.97, np.zeros(3), rollover=True)
    c_c = CO.rollout(s, p, a, 0.97, np.zeros(3), dtype=np.float64, rollover=True)
    assert np.abs(c_np - c_c).max() < 1e-9
    c_off = CO.rollout(s, p, a, 0.97, np.zeros(3), dtype=np.float64)
    assert np.any(np.abs(c_c - c_off) > 1e-3)
    calm = np.clip(R.hover_action(p, 32, np.float64)[None] + 0.05 * rng.normal(size=(N, 32, 4)), -1, 1)
    assert np.array_equal(CO.rollout(s, p, calm, 1.0, np.zeros(3), dtype=np.float64, rollover=True),
                          CO.rollout(s, p, calm, 1.0, np.zeros(3), dtype=np.float64))


# ---- round 3: the other disturbance models (free.py:10-58) and the "tracking_slow" reward (utils.py:297-313) -----------
def test_kat_realworld_reward_and_disturbance_models():
    """Hand-derived from the cited lines (KAT 11-14)."""
    s, p = _rest_state(np.float64)
    # KAT 11: r = -0.02 (5 mean(dp^2) + 3 (1 - w^2))
    assert abs(R.tracking_realworld_reward_fn(s.replace(pos=np.array([1.0, 0, 0]))) - (-1.0 / 30.0)) < 1e-15
    sq = s.replace(quat=np.array([0, 0, np.sin(np.pi / 4), np.cos(np.pi / 4)]))
    assert abs(R.tracking_realworld_reward_fn(sq) - (-0.03)) < 1e-15
    assert abs(R.tracking_realworld_reward_fn(s.replace(pos=np.array([0.3, -0.6, 0.9]), quat=np.array([0.0, 0.6, 0.0, 0.8])))
               - (-(5 * (0.09 + 0.36 + 0.81) / 3 + 3 * 0.36) * 0.02)) < 1e-15
    # KAT 12: drag = -|scale| rel |rel| / 1.5^2, rel = v - 0.5 dp[:3]
    pd = p.replace(disturb_params=(1.0, 0.0, -2.0, 0, 0, 0))
    f = R.drag_disturb(pd, s.replace(vel=np.array([2.0, -3.0, 0.5])))
    assert np.allclose(f, [-0.2 * 1.5 * 1.5 / 2.25, 0.2 * 9 / 2.25, -0.2 * 1.5 * 1.5 / 2.25], atol=1e-15)
    # KAT 13: sin = dp[:3] scale sin(2 pi t / (dp[:3] period/3 + period) + 2 pi dp[3:6])
    ps = p.replace(disturb_params=(1.0, 0.5, 0.0, 0.25, 0.0, 0.0))
    f = R.sin_disturb(ps, s.replace(time=10))
    assert np.allclose(f, [0.2 * np.cos(0.3 * np.pi), 0.1 * np.sin(2 * np.pi * 10 / (50 / 6 + 50)), 0.0], atol=1e-15)
    # KAT 14: periodic holds the state's vector except when time % period == 0
    u = np.array([0.11, -0.07, 0.19])
    sf = s.replace(f_disturb=np.array([0.01, 0.02, 0.03]))
    assert np.array_equal(R.period_disturb(u, p, sf.replace(time=100)), u)
    assert np.array_equal(R.period_disturb(u, p, sf.replace(time=101)), sf.f_disturb)
    # mixed = (drag + sin + periodic) / 3
    sm = sf.replace(time=150, vel=np.array([2.0, -3.0, 0.5]))
    pm = p.replace(disturb_params=(1.0, 0.5, -2.0, 0.25, 0.0, 0.5))
    assert np.allclose(R.mixed_disturb(u, pm, sm), (R.drag_disturb(pm, sm) + R.sin_disturb(pm, sm) + u) / 3, atol=1e-16)
    # deterministic=True only switches the gaussian model off (quadrotor.py:234-235)
    assert np.array_equal(R.disturb_func("gaussian", p, s, np.ones(3), deterministic=True), np.zeros(3))
    assert np.allclose(R.disturb_func("gaussian", p, s, np.ones(3)), 0.05)
    assert np.array_equal(R.disturb_func("drag", pd, s.replace(vel=np.array([2.0, -3.0, 0.5])), None, deterministic=True),
                          R.drag_disturb(pd, s.replace(vel=np.array([2.0, -3.0, 0.5]))))


@pytest.mark.parametrize("kind", ["periodic", "sin", "drag", "mixed"])
@pytest.mark.parametrize("reward", ["penyaw", "realworld"])
def test_c_oracle_disturbance_models_match_numpy(kind, reward):
    """The C rollout with each disturbance model / reward against the literal numpy form, fp64 and fp32; the horizon
    crosses a multiple of disturb_period so that the periodic model both holds and redraws."""
    s, p, rng = make_problem(seed=11, time=35)
    p = p.replace(disturb_params=tuple(float(np.float32(x)) for x in rng.uniform(-1, 1, 6)))
    a = np.clip(R.hover_action(p, 32, np.float64)[None] + 0.3 * rng.normal(size=(24, 32, 4)), -1, 1)
    a = a.astype(np.float32).astype(np.float64)
    draw = rng.uniform(-p.disturb_scale, p.disturb_scale, 3).astype(np.float32).astype(np.float64)
    d = R.Disturb(kind, draw, deterministic=True)
    cost_np, rew_np, poses_np = R.rollout(s, p, a, 0.97, d, reward_fn=R.REWARD_FNS[reward])
    cost_c, rew_c, poses_c = CO.rollout(s, p, a, 0.97, dtype=np.float64, want_rewards=True, want_poses=True, reward=reward,
                                        disturb=d)
    assert np.abs(cost_np - cost_c).max() < 1e-12 and np.abs(rew_np - rew_c).max() < 1e-13
    assert np.abs(poses_np - poses_c).max() < 1e-13
    cost32 = CO.rollout(s.astype(np.float32), p, a, 0.97, dtype=np.float32, reward=reward, disturb=d)
    assert np.abs(cost32 - cost_c).max() < 2e-5 * max(1.0, np.abs(cost_c).max())
    # the model matters: the same rollout without it differs
    cost_0 = CO.rollout(s, p, a, 0.97, dtype=np.float64, reward=reward)
    assert np.abs(cost_0 - cost_c).max() > 1e-6


@pytest.mark.parametrize("kind,reward", [("periodic", "penyaw"), ("sin", "realworld"), ("drag", "penyaw"),
                                         ("mixed", "realworld"), ("none", "realworld")])
def test_c_hessian_with_disturbance_models_matches_torch_ad(kind, reward):
    from oracle import ref_torch as RT
    s, p, rng = make_problem(seed=12, time=40)
    p = p.replace(disturb_params=tuple(float(np.float32(x)) for x in rng.uniform(-1, 1, 6)))
    H = 12
    a = (R.hover_action(p, H, np.float64) + 0.2 * rng.normal(size=(H, 4))).reshape(-1)
    draws = rng.uniform(-p.disturb_scale, p.disturb_scale, (H, 3))
    Rc = CO.hessian(s, p, a, H, reward=reward, kind=kind, draws=draws)
    Rt = RT.hessian(s, p, a, H, reward_kind=reward, kind=kind, draws=draws)
    assert np.abs(Rc - Rt).max() < 1e-10 * max(1.0, np.abs(Rt).max())
    assert np.abs(Rc - Rc.T).max() == 0 and np.all(Rc[-4:] == 0)
    if kind in ("drag", "mixed"):  # the velocity-dependent force enters the second derivatives
        R0 = CO.hessian(s, p, a, H, reward=reward, kind="none")
        assert np.abs(R0 - Rc).max() > 1e-8
    # value check of the objective itself: numpy literal form == torch restatement
    f_np = R.hessian_objective(s, p, a, H, reward_fn=R.REWARD_FNS[reward], kind=kind, draws=draws)
    import torch
    f_t = RT.make_objective(s, p, H, reward, kind, draws)(torch.as_tensor(a)).item()
    assert abs(f_np - f_t) < 1e-12


@pytest.mark.parametrize("kind", ["periodic", "sin", "drag", "mixed"])
def test_force_table_semantics_equal_the_model_functions(kind):
    """The per-step table of include/covo_hip.h (row k = {g_k, c_k}: f_k = c_drag drag(vel_{k-1}) + c_k f_{k-1} + g_k) is
    a re-bracketing of free.py:10-58: the oracle Hessian fed the fp64 table equals the oracle Hessian on the model functions."""
    s, p, rng = make_problem(seed=13, time=44)
    p = p.replace(disturb_params=tuple(float(np.float32(x)) for x in rng.uniform(-1, 1, 6)), disturb_period=6)
    H = 12
    a = (R.hover_action(p, H, np.float64) + 0.2 * rng.normal(size=(H, 4))).reshape(-1)
    draws = rng.uniform(-p.disturb_scale, p.disturb_scale, (H, 3))
    tab = table_ref(p, s, kind, draws, H)
    R1 = CO.hessian(s, p, a, H, kind=kind, draws=draws)
    R2 = CO.hessian(s, p, a, H, kind=kind, table=tab)
    assert np.abs(R1 - R2).max() < 1e-12 * max(1.0, np.abs(R1).max())


def test_jax_bitstream_oracle_known_answers():
    """oracle/jax_rng_np.py -- the independent checker of the product's jax-stream path (covo_randn_jax, noise_stream = "jax") --
    against everything jax publishes about its default stream: the three Random123 threefry2x32-20 vectors jax's own tests use,
    and the values its documentation prints for PRNGKey(0) (split: "JAX PRNG design"; uniform / normal: "Sharp bits"; the ten
    normals of the quickstart) and normal(PRNGKey(42)).  The fp32 erfinv is bounded by scipy's double-precision one."""
    from oracle import jax_rng_np as J
    for key, ctr, exp in [((0, 0), (0, 0), (0x6B200159, 0x99BA4EFE)),
                          ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x1CB996FC, 0xBB002BE7)),
                          ((0x13198A2E, 0x03707344), (0x243F6A88, 0x85A308D3), (0xC4923A9C, 0x483DF7A0))]:
        y0, y1 = J.threefry_block(key[0], key[1], [ctr[0]], [ctr[1]])
        assert (int(y0[0]), int(y1[0])) == exp
    k = J.prng_key(0)
    assert k.tolist() == [0, 0]
    assert J.split(k).tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
    assert abs(float(J.uniform(k, 1)[0]) - 0.41845703) < 1e-8
    assert abs(float(J.normal(k, 1)[0]) - (-0.20584226)) < 2e-8
    quick = [-0.3721109, 0.26423115, -0.18252768, -0.7368197, -0.44030377, -0.1521442, -0.67135346, -0.5908641, 0.73168886,
             0.5673026]
    assert np.abs(J.normal(k, 10) - np.array(quick, dtype=np.float32)).max() < 1e-7
    assert abs(float(J.normal(J.prng_key(42), 1)[0]) - (-0.18471177)) < 2e-8
    z, zx = J.normal(J.prng_key(9), 1 << 18), J.normal_exact(J.prng_key(9), 1 << 18)
    d = np.abs(z - zx)
    assert d.max() < 5e-5 and d[np.abs(zx) < 3].max() < 4e-6 and abs(z.std() - 1) < 5e-3
    # layout: odd counts are padded with one zero counter and cut again; the controllers' row of a sharded draw
    assert np.array_equal(J.bits(k, 3), J._stream(k, [0, 1, 2]))
    e = J.controller_epsilon(J.prng_key(7), 6, offset=2, count=3)
    assert e.shape == (3, 128) and np.array_equal(e[0], J.normal(J.split(J.prng_key(7), 6)[2], 128))
    # and the product's host twin (covo_mpc_amd/random_jax.py: another expression of the same published algorithms) says the same
    from covo_mpc_amd import random_jax as rj
    assert np.array_equal(rj.split(rj.PRNGKey(3), 5), J.split(J.prng_key(3), 5))
    assert np.array_equal(rj.controller_epsilon(rj.PRNGKey(7), 6, sample_offset=2, n_samples=3), e)
    em = J.controller_epsilon_mppi(J.prng_key(5), 16, count=4)
    assert np.array_equal(rj.controller_epsilon_mppi(rj.PRNGKey(5), 16, n_samples=4), em)
