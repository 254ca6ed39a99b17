#!/usr/bin/env python
"""(cost errors of 1.0-1.1e-5 on a sample whose 32 rewards cancel to ~0 are fp32 accumulation: the fp32 C oracle has them too.)
Randomised sweep of the GPU parity tests beyond their fixed parameters (rollout vs the fp64 oracle with position statistics,
ragged sample counts, times near the episode end; Sigma path on the dumped closed-loop Hessians).  Run on the GPU box."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_parity as T

rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
import numpy as np
n_ok = 0
worst_abs = 0.0
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    time = rnd.choice([0, 1, 5, 37, 150, 268, 269, 285, 299, 300, 330])
    N = rnd.choice([1, 63, 64, 65, 200, 257, 1000, 4096, 8192, 16385, 33000, 70000])
    seed = rnd.randrange(1000)
    # test_rollout_vs_fp64_oracle's body with the fp32 error model made explicit: a cost is a sum of 32 rewards of either sign
    # (|r_k| up to ~10 when a sample flies off), so its fp32 error scales with sum_k |r_k|, not with |sum_k r_k| -- the test's
    # rel_err (denominator max(|cost|, 1)) reads 1-3e-5 on the few samples of a large sweep whose rewards cancel
    s_, p_, rng_ = T.make_problem(seed=seed, time=time)
    a = T.sample_actions(p_, rng_, N)
    fs = np.array([0.01, -0.02, 0.03], dtype=np.float32)
    core = T.SamplingCore(N, 32, 0.01, 1.0, device=T.DEV)
    cost = T._run_rollout(core, s_, p_, a, fs, want_stats=True)
    ref, rew, poses = T.CO.rollout(s_, p_, a.astype(np.float64), 1.0, fs.astype(np.float64), dtype=np.float64, want_rewards=True, want_poses=True)
    scale = np.maximum(np.abs(rew).sum(axis=-1), 1.0)
    err = np.abs(cost - ref) / scale
    info = core.info(T.dev_state(s_))
    pm, ps = T.R.pos_stats(poses)
    dm = np.abs(info["pos_mean"].cpu().numpy() - pm).max(); ds = np.abs(info["pos_std"].cpu().numpy() - ps).max()
    bm = core.blockmin.cpu().numpy()
    bm_ok = np.array_equal(bm, np.array([cost[j:j + 64].min() for j in range(0, N, 64)], dtype=np.float32))
    worst_abs = max(worst_abs, float(err.max()))
    if err.max() < 1e-5 and dm < 2e-5 and ds < 2e-5 and bm_ok:
        n_ok += 1
        print("ok rollout", time, seed, N, "test-style rel err %.2e" % T.rel_err(cost, ref).max(), flush=True)
    else:
        print("FAIL rollout", time, seed, N, "err / sum|r|", err.max(), "pos_mean", dm, "pos_std", ds, "blockmin", bm_ok, flush=True)
print("rollout worst |cost - ref| / max(1, sum_k |r_k|):", worst_abs)
for N in (4097, 20000, 66000):
    T.test_rollout_workgroup_shapes(N)
print("rollout:", n_ok, "passed", flush=True)


class _MP:  # pytest's monkeypatch, as far as the tests use it
    def setenv(self, k, v):
        os.environ[k] = v
    def delenv(self, k, raising=False):
        os.environ.pop(k, None)


n2 = 0
for i in range(int(sys.argv[3]) if len(sys.argv) > 3 else 12):
    name = rnd.choice(["covo-online", "mppi", "covo-online"])
    N = rnd.choice([1, 31, 64, 65, 100, 1000, 2047, 4096, 9000, 16384, 40000, 131072])
    lam = rnd.choice(["0.01", "0.1", "1.0"])
    graph = rnd.choice(["graph", "eager"])
    for k in ("COVO_GRAPH", "COVO_NO_GRAPH"):
        os.environ.pop(k, None)
    try:
        T.test_fused_step_ragged_sizes_and_warm_lambda(name, N, lam, graph, _MP())
        n2 += 1
        print("ok fused", name, N, lam, graph, flush=True)
    except AssertionError as e:
        print("FAIL fused", name, N, lam, graph, repr(e)[:300], flush=True)
for k in ("COVO_GRAPH", "COVO_NO_GRAPH"):
    os.environ.pop(k, None)
for N in (33, 129, 200):
    T.test_tiny_and_ragged_sample_counts(N)
for lam, N in ((0.01, 100000), (0.5, 777), (10.0, 65)):
    T.test_softmax_update_vs_oracle(lam, N)
print("fused:", n2, "passed")

# ---- Sigma chain on random spectra: bottom multiplicity, bottom gap, width, exact null spaces; batch 1 (persistent tails) and batched
import numpy as np, torch
nrng = np.random.default_rng(rnd.randrange(1 << 30))
core = T.SamplingCore(256, 32, 0.01, 1.0, device=T.DEV)
worst = 0.0
n3 = 0
for trial in range(int(sys.argv[4]) if len(sys.argv) > 4 else 24):
    Q, _ = np.linalg.qr(nrng.standard_normal((128, 128)))
    lmin = nrng.uniform(-5.0, 0.5)
    width = float(nrng.choice([5.0, 50.0, 500.0, 5000.0]))
    mult = int(nrng.choice([1, 1, 2, 3, 5]))
    gap = float(nrng.choice([1e-9, 1e-6, 1e-3, 0.1, 1.0]))
    w = np.sort(nrng.uniform(lmin + gap, lmin + width, 128))
    w[:mult] = lmin
    if mult < 128 and nrng.random() < 0.5:
        w[mult] = lmin + gap
    Rm = (Q * w) @ Q.T
    if nrng.random() < 0.3:  # exact 4-dim null block like a real CoVO Hessian (last action never reaches a reward)
        Rm[124:, :] = 0
        Rm[:, 124:] = 0
    Rm = 0.5 * (Rm + Rm.T)
    ref = T.R.optimize_sigma(Rm, 0.5, 32, 4)
    for batch in (1, 3):
        Rb = np.stack([Rm] * batch)
        Sigma, L = core.sigma(torch.from_numpy(Rb).to(T.DEV), 0.5, batch=batch)
        S = Sigma[batch - 1].cpu().numpy()
        err = np.linalg.norm(S - ref) / np.linalg.norm(ref)
        worst = max(worst, err)
        L64 = L[batch - 1].cpu().numpy().astype(np.float64)
        errL = np.linalg.norm(L64 @ L64.T - S) / np.linalg.norm(S)
        ok = err < 1e-6 and errL < 2e-7 and np.all(np.isfinite(S))
        if not ok:
            print("FAIL sigma", trial, "batch", batch, "lmin %.3f width %g mult %d gap %g" % (lmin, width, mult, gap), "err", err, "errL", errL, flush=True)
            os.makedirs("gpurun_out", exist_ok=True)
            np.save(f"gpurun_out/fail_sigma_{trial}.npy", Rm)
            for on in (1, 0):  # with / without the deflation (sigma_ns.hip)
                core.lib.covo_debug_set_ns_deflate(core.h, on)
                S2, _ = core.sigma(torch.from_numpy(Rm[None].copy()).to(T.DEV), 0.5)
                e2 = np.linalg.norm(S2[0].cpu().numpy() - ref) / np.linalg.norm(ref)
                print("    deflate", on, "err", e2, "chain (squarings, iterations, deflated)", T._sigma_chain_iters(core), flush=True)
            core.lib.covo_debug_set_ns_deflate(core.h, 1)
        else:
            n3 += 1
print("sigma:", n3, "passed, worst rel err", worst)

# ---- adjoint Hessian against the C oracle's hyper-dual one: random states / times / action scales, clip ties and saturated actions
n4 = 0
worst = 0.0
for trial in range(int(sys.argv[5]) if len(sys.argv) > 5 else 16):
    time = rnd.choice([0, 3, 37, 150, 268, 275, 290, 299, 310])
    s, p, rng = T.make_problem(seed=rnd.randrange(10000), time=time)
    scale = rnd.choice([0.02, 0.1, 0.5, 1.5])
    a = (T.R.hover_action(p, 32, np.float64) + scale * rng.normal(size=(32, 4))).astype(np.float32)
    if rnd.random() < 0.5:
        a[rnd.randrange(32), rnd.randrange(4)] = 1.0   # exact clip tie
        a[rnd.randrange(32), rnd.randrange(4)] = -1.0
    ds = T.dev_state(s)
    Rm = core.hessian(ds.packed, ds, T.EnvParams3D().to_c(), torch.from_numpy(a.reshape(-1)).to(T.DEV))[0].cpu().numpy()
    ref = T.CO.hessian(s, p, a.reshape(-1).astype(np.float64), 32)
    err = np.abs(Rm - ref).max() / max(1.0, np.abs(ref).max())
    worst = max(worst, err)
    if not (err < 1e-9 and np.abs(Rm - Rm.T).max() == 0.0):
        print("FAIL hessian", trial, "time", time, "scale", scale, "err", err, "|ref|", np.abs(ref).max(), flush=True)
    else:
        n4 += 1
print("hessian:", n4, "passed, worst", worst)

# ---- round 3: rollout with random disturbance model / reward / key / period / size against the fp64 oracle with the same draws
import test_gpu_models as M
from covo_mpc_amd import _lib as L_
n5 = 0
worst = 0.0
for trial in range(int(sys.argv[6]) if len(sys.argv) > 6 else 24):
    kind = rnd.choice(M.KINDS)
    reward = rnd.choice(["penyaw", "realworld"])
    time = rnd.choice([0, 7, 37, 49, 50, 150, 268, 290, 299, 305])
    N = rnd.choice([1, 63, 65, 257, 1000, 4096, 16448, 40001])
    period = rnd.choice([1, 3, 7, 50])
    rollover = rnd.random() < 0.3
    s, p, rng = T.make_problem(seed=rnd.randrange(10000), time=time)
    p = p.replace(disturb_params=tuple(float(np.float32(x)) for x in rng.uniform(-1, 1, 6)), disturb_period=period)
    a = T.sample_actions(p, rng, N, sigma=rnd.choice([0.2, 0.5, 1.0]))
    key = M.cr.PRNGKey(rnd.randrange(1 << 30))
    disc = rnd.choice([1.0, 0.97])
    core = T.SamplingCore(N, 32, 0.01, disc, device=T.DEV)
    pc = M.params_c(p, kind, reward, rollover=rollover)
    tab = core.disturb_table(pc, T.dev_state(s).packed, key=key, key_mode=L_.DISTURB_KEYS_SHARED, deterministic=True)
    cost = M._rollout_dev(core, s, pc, a, tab=tab, want_stats=rnd.random() < 0.5)
    draw = M.cr.uniform(M.disturb_key(key), (3,), -p.disturb_scale, p.disturb_scale).astype(np.float64)
    d = T.R.Disturb(kind, draw, True)
    ref = T.CO.rollout(s, p, a.astype(np.float64), disc, dtype=np.float64, rollover=rollover, reward=reward, disturb=d)
    ref32 = T.CO.rollout(s.astype(np.float32), p, a, disc, dtype=np.float32, rollover=rollover, reward=reward, disturb=d)
    err = np.minimum(T.rel_err(cost, ref), T.rel_err(cost, ref32.astype(np.float64)))
    worst = max(worst, float(np.median(err)))
    if (err < 1.3e-5).mean() > 0.995:
        n5 += 1
    else:
        print("FAIL model rollout", kind, reward, time, N, period, rollover, disc, "max err", err.max(), "frac ok", (err < 1.3e-5).mean(), flush=True)
print("model rollouts:", n5, "passed, worst median err", worst)

# ---- round 3: adjoint Hessian (13- and 16-component instantiations) with random disturbance model / reward / period / time / keys
n6 = 0
worst = 0.0
hcore = T.SamplingCore(256, 32, 0.01, 1.0, device=T.DEV)
for trial in range(int(sys.argv[7]) if len(sys.argv) > 7 else 16):
    kind = rnd.choice(M.KINDS + ["none"])
    reward = rnd.choice(["penyaw", "realworld"])
    time = rnd.choice([0, 3, 37, 49, 150, 268, 275, 290, 299])
    period = rnd.choice([1, 3, 7, 50])
    s, p, rng = T.make_problem(seed=rnd.randrange(10000), time=time)
    p = p.replace(disturb_params=tuple(float(np.float32(x)) for x in rng.uniform(-1, 1, 6)), disturb_period=period)
    a = (T.R.hover_action(p, 32, np.float64) + rnd.choice([0.02, 0.1, 0.5]) * rng.normal(size=(32, 4))).astype(np.float32)
    if rnd.random() < 0.5:
        a[rnd.randrange(32), rnd.randrange(4)] = 1.0
    key = M.cr.PRNGKey(rnd.randrange(1 << 30))
    ds = T.dev_state(s)
    pc = M.params_c(p, kind, reward)
    tab = None
    if kind in M.KINDS:
        tab = hcore.disturb_table(pc, ds.packed, key=key, key_mode=L_.DISTURB_KEYS_HESSIAN, deterministic=True)
    Rm = hcore.hessian(ds.packed, ds, pc, torch.from_numpy(a.reshape(-1)).to(T.DEV), f_steps=tab)[0].cpu().numpy()
    Rp = hcore.hessian(ds.packed, ds, pc, torch.from_numpy(a.reshape(-1)).to(T.DEV), method="pairs", f_steps=tab)[0].cpu().numpy()
    ref = T.CO.hessian(s, p, a.reshape(-1).astype(np.float64), 32, reward=reward, kind=kind,
                       table=None if tab is None else tab[0].cpu().numpy())
    sc = max(1.0, np.abs(ref).max())
    err, errp = np.abs(Rm - ref).max() / sc, np.abs(Rp - ref).max() / sc
    worst = max(worst, err, errp)
    if np.all(np.isfinite(ref)) and err < 1e-9 and errp < 1e-9 and np.abs(Rm - Rm.T).max() == 0.0:
        n6 += 1
    elif not np.all(np.isfinite(ref)):  # norm'(0) = NaN in the reference's AD: both must be NaN at the same places
        n6 += int(np.array_equal(np.isnan(ref), np.isnan(Rm)))
    else:
        print("FAIL model hessian", kind, reward, time, period, "adjoint", err, "pairs", errp, flush=True)
print("model hessians:", n6, "passed, worst", worst)
