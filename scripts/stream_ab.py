#!/usr/bin/env python
"""A/B of covo-online's noise GEMM streamed inside the Sigma chain's finalize launch against the GEMM as its own launch: graph
replays of the step's launch groups (covo_debug_time_step), same state, same box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr, _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="gaussian", disable_rollover_terminate=True,
                     generate_noisy_state=True, device="cuda:0")
params = env.default_params
lib = _lib.load_library()
for rnd in range(2):
    for on in (1, 0):
        c, _ = cm.envs.get_controller(env, "covo-online", f"N{N}_H32_lam0.01", device="cuda:0", compute_info=False)
        _lib.check(lib.covo_debug_set_stream_gemm(c.core.h, on), "stream")
        obs, info, state = env.reset(cr.PRNGKey(1), params)
        cp = c.reset(state, params, c.init_control_params, cr.PRNGKey(2))
        key = cr.PRNGKey(3)
        for step in range(30):
            key, k_act, k_step = cr.split(key, 3)
            u, cp, _ = c(obs, state, params, k_act, cp, info)
            obs, state, reward, done, info = env.step(k_step, state, u.cpu().numpy(), params)
        T = c.core.time_phases
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ds = info["noisy_state"]
        e0.record()
        for i in range(100):
            u, cp, _ = c(obs, state, params, k_act, cp, info)
        e1.record(); torch.cuda.synchronize()
        print(f"stream={on}: sigma {T(4):.1f}  sigma+gemm {T(12):.1f}  +rollout {T(28):.1f}  whole {T(63):.1f} us; 100 real steps on one state: "
              f"{e0.elapsed_time(e1) * 10:.1f} us/step")
        c.core.close()
