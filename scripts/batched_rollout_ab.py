#!/usr/bin/env python
"""A/B of the env-batched rollout launch (BASELINE configs[4]: 32 instances x N = 4 096 in one launch, softmax records on) against
the plain kernel on as many samples in ONE instance (N = 131 072), with and without the record epilogue -- what does BATCHED add?
(VERDICT r05 Next 3.)  usage: python scripts/batched_rollout_ab.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covo_mpc_amd as cm  # noqa: E402
from covo_mpc_amd import random as cr  # noqa: E402
from covo_mpc_amd.controllers._core import SamplingCore  # noqa: E402

dev = "cuda:0"
E, N = 32, 4096
env = cm.envs.Quad3D(task="tracking", obs_type="quad_params", enable_randomizer=True, disturb_type="gaussian",
                     disable_rollover_terminate=True, generate_noisy_state=True, device=dev)
c0, _ = cm.envs.get_controller(env, "covo-online", f"N{N}_H32_lam0.01", device=dev, compute_info=False)
cp0 = c0.init_control_params
c0.core.close()
params = [env.sample_params(cr.PRNGKey(1000 + g)) for g in range(E)]
b = cm.controllers.BatchedCoVOController(env, E, N, 32, 0.01, discount=cp0.discount, gamma_mean=cp0.gamma_mean,
                                         sample_sigma=cp0.sample_sigma, a_mean_init=cp0.a_mean, device=dev)
ep = cm.envs.BatchedDeviceEpisode(env, [cr.PRNGKey(2000 + g) for g in range(E)], params, (b.core.lib, b.core.h), dev)
rngs = np.stack([np.asarray(cr.PRNGKey(3000 + g)) for g in range(E)])
rngs = b.run_episode(ep, rngs, 20)
torch.cuda.synchronize()
keys = np.stack([np.asarray(cr.PRNGKey(4000 + g)) for g in range(E)]).astype(np.uint32)
for _ in range(3):
    keys[:, 1] += 1
    b(None, keys)
torch.cuda.synchronize()
print(f"batched {E} x {N}: rollout launch (records on) in a graph of 10 copies: {min(b.time_phases(16) for _ in range(3)):.2f} us; "
      f"GEMM -> rollout pair minus GEMM: {min(b.time_phases(8 | 16) for _ in range(3)) - min(b.time_phases(8) for _ in range(3)):.2f} us")
b.core.close()
# the plain kernel on one instance with as many samples
envp = cm.envs.Quad3D(task="tracking", enable_randomizer=False, disturb_type="gaussian", disable_rollover_terminate=True,
                      generate_noisy_state=True, device=dev)
p0 = envp.default_params
obs, info, state = envp.reset(cr.PRNGKey(1), p0)
ds = info["noisy_state"].to_device(dev)
pc = p0.to_c()
for n in (65536, 131072, 262144):
    core = SamplingCore(n, 32, 0.01, 1.0, device=dev, compute_info=False, trust_clipped=True, use_graph=False)
    am = torch.tensor([-0.3378, 0.0, 0.0, 0.0], device=dev).repeat(32)
    core.noise_gemm_philox((0.5 * torch.eye(128, device=dev)).contiguous(), am, (1, 2))
    torch.cuda.synchronize()
    r0 = core.time_rollout(ds, pc, reps=100, with_records=False)
    r1 = core.time_rollout(ds, pc, reps=100, with_records=True)
    print(f"plain N={n}: without records {r0[0]:.2f} us (fastest batch {r0[1]:.2f}); with records {r1[0]:.2f} us ({r1[1]:.2f}); "
          f"per 131 072 samples: {r0[0] * 131072 / n:.2f} / {r1[0] * 131072 / n:.2f}")
    core.close()


def live_stats(cost, lam=0.01, chunk=256):
    c = cost.reshape(-1, chunk)
    m = c.min(axis=1, keepdims=True)
    live = ((c - m) / lam < 103.0).sum(axis=1)
    return float(live.mean()), int(live.max())


# how many samples carry a non-zero weight against their workgroup's LOCAL minimum (the record epilogue re-reads their stripes)
b = cm.controllers.BatchedCoVOController(env, E, N, 32, 0.01, discount=cp0.discount, gamma_mean=cp0.gamma_mean,
                                         sample_sigma=cp0.sample_sigma, a_mean_init=cp0.a_mean, device=dev)
ep = cm.envs.BatchedDeviceEpisode(env, [cr.PRNGKey(2000 + g) for g in range(E)], params, (b.core.lib, b.core.h), dev)
rngs = np.stack([np.asarray(cr.PRNGKey(3000 + g)) for g in range(E)])
rngs = b.run_episode(ep, rngs, 21)
torch.cuda.synchronize()
print("batched step: live samples per 256-sample workgroup (mean, max):", live_stats(b._cost.cpu().numpy()))
b.core.close()
core = SamplingCore(131072, 32, 0.01, 1.0, device=dev, compute_info=False, trust_clipped=True, use_graph=False)
am = torch.tensor([-0.3378, 0.0, 0.0, 0.0], device=dev).repeat(32)
core.noise_gemm_philox((0.5 * torch.eye(128, device=dev)).contiguous(), am, (1, 2))
core.rollout(ds, pc, (0.0, 0.0, 0.0), False)
torch.cuda.synchronize()
print("plain N=131072 (reset state, sigma 0.5 I): live samples per workgroup (mean, max):", live_stats(core.cost.cpu().numpy()))
core.close()
