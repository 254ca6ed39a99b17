#!/usr/bin/env python
"""Dump Hessians for offline spectrum analysis: every 6th step of a closed-loop covo-online episode (two tasks) and of the
bench's teacher-forced PID-tracked episode (bench.py: make_states)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr
import bench
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
        if i % 6 == 0 or (120 <= i < 140):
            ds = info["noisy_state"].to_device(dev)
            am = c.core.shift_mean(cp.a_mean.reshape(-1).contiguous())
            Rs.append(c.core.hessian(ds.packed, ds, c._params_c(params), am)[0].cpu().numpy())
        u, cp, _ = c(obs, state, params, ka, cp, info)
        obs, state, _, _, info = env.step(ks, state, u.cpu().numpy(), params)
    res[task] = np.stack(Rs)
# bench-like teacher-forced sequence
env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="gaussian",
                     disable_rollover_terminate=True, generate_noisy_state=True, device=dev)
params = env.default_params
c, cp = cm.envs.get_controller(env, "covo-online", "N8192_H32_lam0.01", device=dev, compute_info=False)
state0, packed, host_states = bench.make_states(env, params, 300, seed=1)
obs0, info0, s_reset = env.reset(cr.PRNGKey(1), params)
from covo_mpc_amd.dynamics.dataclass import DeviceState
packed_d = torch.from_numpy(packed).to(dev)
dref = s_reset.to_device(dev)
key = cr.PRNGKey(1)
Rs = []
for i in range(300):
    key, k = cr.split(key)
    ds = DeviceState(packed=packed_d[i], pos_traj=dref.pos_traj, vel_traj=dref.vel_traj, time=int(host_states[i].time))
    if i % 6 == 0 or (120 <= i < 140):
        am = c.core.shift_mean(cp.a_mean.reshape(-1).contiguous())
        Rs.append(c.core.hessian(ds.packed, ds, c._params_c(params), am)[0].cpu().numpy())
    u, cp, _ = c(None, None, params, k, cp, {"noisy_state": ds})
res["bench_teacher_forced"] = np.stack(Rs)
np.savez_compressed(out, **res)
print("saved", {k: v.shape for k, v in res.items()})
