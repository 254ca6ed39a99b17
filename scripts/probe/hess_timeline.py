#!/usr/bin/env python
"""Timeline of the merged Hessian launch (hessian_adj_body.hpp: adj_all_kernel, built with -DADJ_STAMPS:
`make -C covo_mpc_amd/csrc VARIANT=stamps HIPFLAGS+=-DADJ_STAMPS`, run with COVO_HIP_LIB pointing at libcovo_hip_stamps.so): the 100 MHz
wall-clock stamps of a few covo-online steps, in microseconds from KB workgroup 0's entry."""
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
WS_STAMPS = 512 + 32 * 13 * 17 + 512 + 33 * 16 + 32 * 256 + 32 * 64 + 32 * 16 + 32 * 16 * 128 + 2 + 128  # adj13
for step in range(12):
    key, k_act, k_step = cr.split(key, 3)
    u, cp, _ = c(obs, state, params, k_act, cp, info)
    obs, state, reward, done, info = env.step(k_step, state, u.cpu().numpy(), params)
    if step < 9:
        continue
    out = torch.zeros(256, dtype=torch.float64).pin_memory()
    _lib.check(c.core.lib.covo_debug_hess_workspace(c.core.h, _lib.ptr(out), WS_STAMPS, 256, c.core.stream()), "ws")
    torch.cuda.synchronize()
    t = out.numpy()
    us = lambda x: (x - t[250]) / 100.0
    kb = us(t[0:32])
    print(f"step {step}: KB waves end {kb.min():.2f} .. {kb.max():.2f}")
    for w in range(9):
        o = 32 + 4 * w
        print(f"   chain {w}: entry {us(t[o + 3]):.2f} flags seen {us(t[o]):.2f} LDS filled {us(t[o + 1]):.2f} end {us(t[o + 2]):.2f}")
    hs, he = us(t[72:136:2]), us(t[73:136:2])
    print(f"   hyper-dual: flags seen {hs.min():.2f} .. {hs.max():.2f}; end {he.min():.2f} .. {he.max():.2f} (last: step {int(np.argmax(he))})")
    ke, ks, kd = us(t[138:244:3]), us(t[136:244:3]), us(t[137:244:3])
    print(f"   KD tiles: entry {ke.min():.2f} .. {ke.max():.2f}; flags seen {ks.min():.2f} .. {ks.max():.2f}; end {kd.min():.2f} .. {kd.max():.2f}")
