#!/usr/bin/env python
"""Every non-headline path once under `rocprofv3 --kernel-trace --stats` (run on the GPU box):
  rocprofv3 --kernel-trace --stats -d gpurun_out/paths -o paths -- python3 scripts/profile_paths.py
mppi (plain and with covariance adaptation), covo-offline, covo-online with the table-driven disturbance models and the
tracking_slow reward, the closed-loop episode driver, the env-batched step.  The point is the per-kernel average of kernels the
headline bench never launches (a one-lane table kernel hid there at 25-43 us for a whole round)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr

DEV = "cuda:0"
N = int(os.environ.get("N", 65536))


def loop(name, task, kind, steps=12, **kw):
    env = cm.envs.Quad3D(task=task, enable_randomizer=False, disturb_type=kind, disable_rollover_terminate=True,
                         generate_noisy_state=True, device=DEV)
    params = env.default_params
    c, cp = cm.envs.get_controller(env, name, f"N{N}_H32_lam0.01", device=DEV, **kw)
    obs, info, state = env.reset(cr.PRNGKey(1), params)
    cp = c.reset(state, params, cp, cr.PRNGKey(3))
    key = cr.PRNGKey(2)
    for i in range(steps):
        key, k, ks = cr.split(key, 3)
        u, cp, _ = c(obs, state, params, k, cp, info)
        obs, state, _, _, info = env.step(ks, state, u.cpu().numpy(), params)
    torch.cuda.synchronize()
    return env, c, cp


loop("mppi", "tracking_zigzag", "gaussian")
env, c, cp = loop("mppi", "tracking_zigzag", "periodic")
loop("covo-offline", "tracking_zigzag", "gaussian")
loop("covo-online", "tracking_slow", "mixed")
loop("covo-online", "tracking_zigzag", "sin", compute_info=True)
# MPPI with covariance adaptation (mppi.py:119-125)
env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="gaussian", disable_rollover_terminate=True,
                     generate_noisy_state=True, device=DEV)
c, cp = cm.envs.get_controller(env, "mppi", f"N{N}_H32_lam0.01", device=DEV)
cp = cp.replace(gamma_sigma=0.5)
c2 = cm.controllers.MPPIController(env, cp, N, 32, 0.01, device=DEV)
obs, info, state = env.reset(cr.PRNGKey(1), env.default_params)
key = cr.PRNGKey(2)
for i in range(12):
    key, k, ks = cr.split(key, 3)
    u, cp, _ = c2(obs, state, env.default_params, k, cp, info)
    obs, state, _, _, info = env.step(ks, state, u.cpu().numpy(), env.default_params)
torch.cuda.synchronize()
# closed-loop episode driver with a table model
env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="periodic", disable_rollover_terminate=True,
                     generate_noisy_state=True, device=DEV)
c, cp = cm.envs.get_controller(env, "covo-online", f"N{N}_H32_lam0.01", device=DEV, compute_info=False)
ep = cm.envs.DeviceEpisode(env, cr.PRNGKey(20), env.default_params, (c.core.lib, c.core.h), c.core.device)
cp = c.reset(ep.state0, env.default_params, c.init_control_params, cr.PRNGKey(22))
cp, _ = c.run_episode(ep, env.default_params, cp, cr.PRNGKey(19), 30)
ep.read_log()
# env-batched step with a table model and domain randomisation
E, Nb = 8, 4096
env = cm.envs.Quad3D(task="tracking", obs_type="quad_params", enable_randomizer=True, disturb_type="mixed",
                     disable_rollover_terminate=True, generate_noisy_state=True, device=DEV)
states, pars, infos = [], [], []
for e in range(E):
    p = env.sample_params(cr.PRNGKey(100 + e))
    obs, info, state = env.reset(cr.PRNGKey(200 + e), p)
    states.append(state); pars.append(p); infos.append(info)
b = cm.controllers.BatchedCoVOController(env, E, Nb, 32, 0.01, device=DEV)
b.set_instances(states, pars)
for i in range(8):
    keys = np.stack([np.asarray(cr.PRNGKey(1000 + 10 * i + e)) for e in range(E)])
    b([inf["noisy_state"] for inf in infos], keys)
torch.cuda.synchronize()
print("done")
