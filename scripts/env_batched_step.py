#!/usr/bin/env python
"""BASELINE configs[4] on one GPU through covo_mpc_step_batched: E env instances (lissajous `tracking`, domain-randomised
parameters) x N samples, ONE graph with a single batched Hessian + Sigma launch set.  Prints control steps/s (all
instances advance one control step per call) and env-steps/s; compare scripts/env_batch.py (one graph per instance)."""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, nargs="+", default=[1, 8, 32, 64])
ap.add_argument("--N", type=int, default=4096)
ap.add_argument("--steps", type=int, default=200)
args = ap.parse_args()
dev = "cuda:0"
env = cm.envs.Quad3D(task="tracking", obs_type="quad_params", enable_randomizer=True, disturb_type="gaussian",
                     disable_rollover_terminate=True, generate_noisy_state=True, device=dev)
c0, cp0 = cm.envs.get_controller(env, "covo-online", f"N{args.N}_H32_lam0.01", device=dev, compute_info=False)
cp0 = c0.init_control_params
for E in args.envs:
    states, params, noisy = [], [], []
    for e in range(E):
        p = env.sample_params(cr.PRNGKey(100 + e))
        obs, info, st = env.reset(cr.PRNGKey(200 + e), p)
        states.append(st); params.append(p); noisy.append(info["noisy_state"])
    b = cm.controllers.BatchedCoVOController(env, E, args.N, 32, 0.01, discount=cp0.discount, gamma_mean=cp0.gamma_mean,
                                             sample_sigma=cp0.sample_sigma, a_mean_init=cp0.a_mean, device=dev)
    b.set_instances(states, params)
    packed = torch.stack([s.to_device(dev).packed for s in noisy])
    keys = np.stack([np.asarray(cr.PRNGKey(300 + e)) for e in range(E)])
    for _ in range(5):
        b(packed, keys)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        keys[:, 1] += 1
        b(packed, keys)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"batched covo-online E={E} N={args.N}: {1e6 * el / args.steps:.0f} us per call = {E * args.steps / el:.0f} env-steps/s "
          f"({1e6 * el / args.steps / E:.1f} us per instance; host enqueue {1e6 * th / args.steps:.0f} us per call)", flush=True)
    del b
