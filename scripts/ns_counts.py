#!/usr/bin/env python
"""Per control step: how many squarings and Newton-Schulz iterations the Sigma chain actually runs on the Hessians of
a closed-loop covo-online episode (tracking_zigzag) -- the input of the adaptive launch-count policy (step.hip)."""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covo_mpc_amd as cm
from covo_mpc_amd import _lib, random as cr
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for task in ("tracking_zigzag", "tracking"):
    env = cm.envs.Quad3D(task=task, enable_randomizer=False, disturb_type="gaussian",
                         disable_rollover_terminate=True, generate_noisy_state=True, device=dev)
    c, _ = cm.envs.get_controller(env, "covo-online", f"N{N}_H32_lam0.01", device=dev, compute_info=False)
    params = env.default_params
    obs, info, state = env.reset(cr.PRNGKey(1), params)
    cp = c.reset(state, params, c.init_control_params, cr.PRNGKey(2))
    key = cr.PRNGKey(3)
    SC0 = 11 * 128 * 128
    out_t = torch.zeros(16, dtype=torch.float64).pin_memory()
    sq, it = [], []
    for i in range(steps):
        key, ka, ks = cr.split(key, 3)
        u, cp, _ = c(obs, state, params, ka, cp, info)
        _lib.check(c.core.lib.covo_debug_sigma_workspace(c.core.h, _lib.ptr(out_t), SC0, 16, c.core.stream()))
        torch.cuda.synchronize()
        sq.append(int(out_t[7])); it.append(int(out_t[6]))
        obs, state, _, _, info = env.step(ks, state, u.cpu().numpy(), params)
    print(task, "squarings:", sorted(collections.Counter(sq).items()), " NS iterations:", sorted(collections.Counter(it).items()))
    print("  first 60 squarings:", sq[:60])
    print("  changes step-to-step: squarings", int(np.sum(np.diff(sq) != 0)), " NS", int(np.sum(np.diff(it) != 0)), "of", steps - 1)
