#!/usr/bin/env python
"""Closed-loop soak over the whole model matrix (run on the GPU box): controller x task x disturb_type, one 300-step episode
each through covo_run_episode (control step + env step on the device).  Reports the mean position error, whether every logged
value is finite, the device status and the steps/s -- the matrix the reference's `--controller X --task Y --disturb_type Z`
command line spans (quadrotor.py:755-766)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr

DEV = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
bad = 0
print("%-13s %-16s %-9s %9s %9s %8s" % ("controller", "task", "disturb", "err_pos", "reward", "steps/s"))
for name in ("covo-online", "covo-offline", "mppi"):
    for task in ("tracking_zigzag", "tracking_slow", "hovering"):
        for kind in ("none", "gaussian", "periodic", "sin", "drag", "mixed"):
            env = cm.envs.Quad3D(task=task, enable_randomizer=False, disturb_type=kind, disable_rollover_terminate=True,
                                 generate_noisy_state=True, device=DEV)
            params = env.default_params
            c, cp = cm.envs.get_controller(env, name, f"N{N}_H32_lam0.01", device=DEV, compute_info=False)
            c.alias_outputs = True
            ep = cm.envs.DeviceEpisode(env, cr.PRNGKey(20), params, (c.core.lib, c.core.h), c.core.device)
            cp = c.reset(ep.state0, params, c.init_control_params, cr.PRNGKey(22))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cp, key = c.run_episode(ep, params, cp, cr.PRNGKey(19), params.max_steps_in_episode)
            log = ep.read_log()
            dt = time.perf_counter() - t0
            ok = np.all(np.isfinite(log)) and bool(torch.isfinite(cp.a_mean).all()) and c.core.device_status() == 0
            bad += 0 if ok else 1
            print("%-13s %-16s %-9s %9.4f %9.3f %8.0f %s" % (name, task, kind, log[:, 1].mean(), log[:, 0].mean(), len(log) / dt,
                                                            "" if ok else "NOT FINITE / status"), flush=True)
            c.core.close()
print("bad:", bad)
sys.exit(1 if bad else 0)
