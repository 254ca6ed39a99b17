#!/usr/bin/env python
"""Dump the Hessians of a closed-loop covo-online episode (every k-th step) for offline spectrum analysis."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr
dev = "cuda:0"
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/hessians.npz"
res = {}
for task in ("tracking_zigzag", "tracking"):
    env = cm.envs.Quad3D(task=task, enable_randomizer=False, disturb_type="gaussian",
                         disable_rollover_terminate=True, generate_noisy_state=True, device=dev)
    c, _ = cm.envs.get_controller(env, "covo-online", "N8192_H32_lam0.01", device=dev, compute_info=False)
    params = env.default_params
    obs, info, state = env.reset(cr.PRNGKey(1), params)
    cp = c.reset(state, params, c.init_control_params, cr.PRNGKey(2))
    key = cr.PRNGKey(3)
    Rs = []
    for i in range(300):
        key, ka, ks = cr.split(key, 3)
        if i % 6 == 0:
            ds = info["noisy_state"].to_device(dev)
            am = c.core.shift_mean(cp.a_mean.reshape(-1).contiguous())
            Rs.append(c.core.hessian(ds.packed, ds, params.to_c(), am)[0].cpu().numpy())
        u, cp, _ = c(obs, state, params, ka, cp, info)
        obs, state, _, _, info = env.step(ks, state, u.cpu().numpy(), params)
    res[task] = np.stack(Rs)
np.savez_compressed(out, **res)
print("saved", {k: v.shape for k, v in res.items()})
