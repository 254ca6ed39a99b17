#!/usr/bin/env python
"""GPU time of the batched Hessian (hessian_adj.hip) and Sigma chain (sigma_ns.hip) against the batch size:
real CoVO Hessians of `batch` domain-randomised env instances (covo-offline's table and the env-batched step
run these with batch = 300 / E)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr
dev = "cuda:0"
env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="gaussian",
                     disable_rollover_terminate=True, generate_noisy_state=True, device=dev)
c, _ = cm.envs.get_controller(env, "covo-online", "N1024_H32_lam0.01", device=dev, compute_info=False)
params = env.default_params
obs, info, state = env.reset(cr.PRNGKey(1), params)
core = c.core
ds = state.to_device(dev)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
for B in (1, 4, 16, 32, 64, 300):
    packed = ds.packed.repeat(B, 1).contiguous()
    am = (c.init_control_params.a_mean.reshape(1, 128) + 0.05 * torch.randn(B, 128, device=dev)).contiguous()
    th = timed(lambda: core.hessian(packed, ds, params.to_c(), am, batch=B))
    R = core.hessian(packed, ds, params.to_c(), am, batch=B)
    ts = timed(lambda: core.sigma(R, 0.5, batch=B))
    print(f"batch {B:4d}: hessian {th:8.1f} us ({th / B:6.1f}/matrix)   sigma {ts:8.1f} us ({ts / B:6.1f}/matrix)", flush=True)
