#!/usr/bin/env python
"""Soak of the Sigma chain's persistent launches (run on the GPU box): many closed-loop episodes of covo-online, single
(covo_run_episode) and env-batched (covo_run_episode_batched), checking after every episode that the device status is clean
(no grid-barrier / evaluation time-out) and every logged value finite.  usage: soak_sigma_chain.py [episodes, default 40]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr

DEV = "cuda:0"
n_ep = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="gaussian", disable_rollover_terminate=True,
                     generate_noisy_state=True, device=DEV)
params = env.default_params
c, cp = cm.envs.get_controller(env, "covo-online", "N4096_H32_lam0.01", device=DEV, compute_info=False)
c.alias_outputs = True
t0 = time.perf_counter()
for j in range(n_ep):
    ep = cm.envs.DeviceEpisode(env, cr.PRNGKey(100 + j), params, (c.core.lib, c.core.h), c.core.device)
    cp = c.reset(ep.state0, params, c.init_control_params, cr.PRNGKey(200 + j))
    cp, key = c.run_episode(ep, params, cp, cr.PRNGKey(300 + j), params.max_steps_in_episode)
    log = ep.read_log()
    ok = np.all(np.isfinite(log)) and c.core.device_status() == 0
    bad += 0 if ok else 1
print(f"single: {n_ep} episodes x {params.max_steps_in_episode} steps, bad {bad}, {time.perf_counter() - t0:.1f} s", flush=True)
E, N = 32, 4096
envb = cm.envs.Quad3D(task="tracking", obs_type="quad_params", enable_randomizer=True, disturb_type="gaussian",
                      disable_rollover_terminate=True, generate_noisy_state=True, device=DEV)
pb = [envb.sample_params(cr.PRNGKey(1000 + g)) for g in range(E)]
cp0 = c.init_control_params
b = cm.controllers.BatchedCoVOController(envb, E, N, 32, 0.01, discount=cp0.discount, gamma_mean=cp0.gamma_mean,
                                         sample_sigma=cp0.sample_sigma, a_mean_init=cp0.a_mean, device=DEV)
badb = 0
t0 = time.perf_counter()
for j in range(n_ep):
    ep2 = cm.envs.BatchedDeviceEpisode(envb, [cr.PRNGKey(5000 + g + 100 * j) for g in range(E)], pb, (b.core.lib, b.core.h), DEV)
    b.a_mean.copy_(torch.as_tensor(cp0.a_mean, device=DEV).reshape(1, -1).expand(E, -1))
    r2 = np.stack([np.asarray(cr.PRNGKey(6000 + g + 100 * j)) for g in range(E)])
    b.run_episode(ep2, r2, 300)
    log = ep2.read_log()
    ok = np.all(np.isfinite(log)) and b.core.device_status() == 0 and bool(torch.isfinite(b.a_cov).all())
    badb += 0 if ok else 1
print(f"batched: {n_ep} episodes x 300 steps x {E} instances, bad {badb}, {time.perf_counter() - t0:.1f} s")
sys.exit(1 if bad or badb else 0)
