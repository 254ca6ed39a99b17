#!/usr/bin/env python
"""(cost errors of 1.0-1.1e-5 on a sample whose 32 rewards cancel to ~0 are fp32 accumulation: the fp32 C oracle has them too.)
Randomised sweep of the GPU parity tests beyond their fixed parameters (rollout vs the fp64 oracle with position statistics,
ragged sample counts, times near the episode end; Sigma path on the dumped closed-loop Hessians).  Run on the GPU box."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_parity as T

rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_ok = 0
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    time = rnd.choice([0, 1, 5, 37, 150, 268, 269, 285, 299, 300, 330])
    N = rnd.choice([1, 63, 64, 65, 200, 257, 1000, 4096, 8192, 16385, 33000, 70000])
    seed = rnd.randrange(1000)
    try:
        T.test_rollout_vs_fp64_oracle(time, seed, N)
        n_ok += 1
        print("ok rollout", time, seed, N, flush=True)
    except AssertionError as e:
        print("FAIL rollout", time, seed, N, flush=True)
        import numpy as np
        s_, p_, rng_ = T.make_problem(seed=seed, time=time)
        a = T.sample_actions(p_, rng_, N)
        fs = np.array([0.01, -0.02, 0.03], dtype=np.float32)
        core = T.SamplingCore(N, 32, 0.01, 1.0, device=T.DEV)
        cost = T._run_rollout(core, s_, p_, a, fs, want_stats=True)
        ref, rew, poses = T.CO.rollout(s_, p_, a.astype(np.float64), 1.0, fs.astype(np.float64), dtype=np.float64, want_rewards=True, want_poses=True)
        info = core.info(T.dev_state(s_))
        pm, ps = T.R.pos_stats(poses)
        dm = np.abs(info["pos_mean"].cpu().numpy() - pm); ds = np.abs(info["pos_std"].cpu().numpy() - ps)
        print("  cost rel", T.rel_err(cost, ref).max(), "mean err max", dm.max(), "at", np.unravel_index(dm.argmax(), dm.shape), "std err max", ds.max(), "at", np.unravel_index(ds.argmax(), ds.shape))
        k = np.unravel_index(ds.argmax(), ds.shape)[0]
        print("  std dev/ref at worst step", info["pos_std"].cpu().numpy()[k], ps[k], " step0:", info["pos_std"].cpu().numpy()[0], ps[0])
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
