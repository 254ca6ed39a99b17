#!/usr/bin/env python
"""Timeline of the streamed finalize launch (sigma_ns.hip: ns_finalize_stream_kernel): its 100 MHz wall-clock stamps after a few
covo-online steps at N (default 65536), in microseconds from the factoring workgroup's entry."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr, _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="gaussian", disable_rollover_terminate=True,
                     generate_noisy_state=True, device="cuda:0")
c, _ = cm.envs.get_controller(env, "covo-online", f"N{N}_H32_lam0.01", device="cuda:0", compute_info=False)
params = env.default_params
obs, info, state = env.reset(cr.PRNGKey(1), params)
cp = c.reset(state, params, c.init_control_params, cr.PRNGKey(2))
key = cr.PRNGKey(3)
SC_STAMPS = 384 + 17 * 64 + 12 * 64 + 128 * 8 + 64 + 128 + 64
for step in range(12):
    key, k_act, k_step = cr.split(key, 3)
    u, cp, _ = c(obs, state, params, k_act, cp, info)
    obs, state, reward, done, info = env.step(k_step, state, u.cpu().numpy(), params)
    if step < 8:
        continue
    out = torch.zeros(96, dtype=torch.float64).pin_memory()
    _lib.check(c.core.lib.covo_debug_sigma_workspace(c.core.h, _lib.ptr(out), 11 * 128 * 128 + SC_STAMPS, 96, c.core.stream()), "ws")
    torch.cuda.synchronize()
    t = out.numpy()
    us = lambda x: (x - t[0]) / 100.0
    print(f"step {step}: Z in LDS {us(t[1]):.2f} | panels final", " ".join(f"{us(t[2 + p]):.2f}" for p in range(8)),
          f"| chol end {us(t[10]):.2f} | flags", " ".join(f"{us(t[11 + p]):.2f}" for p in range(8)))
    for name, o in (("first worker", 24), ("last worker", 56)):
        print(f"   {name}: entry {us(t[o]):.2f};", " ; ".join(
            f"p{p}: {us(t[o + 1 + 3 * p]):.2f} {us(t[o + 2 + 3 * p]):.2f} {us(t[o + 3 + 3 * p]):.2f}" for p in range(8)), "(flag seen, staged, done)")
