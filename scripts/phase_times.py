#!/usr/bin/env python
"""Per-phase GPU time of one covo-online control step (graph replay, no profiler): bench.py's workload."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
device = "cuda:0"
env = cm.envs.Quad3D(task="tracking_zigzag", obs_type="quad", enable_randomizer=False, disturb_type="gaussian",
                     disable_rollover_terminate=True, generate_noisy_state=True, device=device)
params = env.default_params
controller, cp = cm.envs.get_controller(env, "covo-online", f"N{N}_H32_lam0.01", device=device, compute_info=False)
state0, packed, host_states = bench.make_states(env, params, 60, seed=1)
obs0, info0, s_reset = env.reset(cr.PRNGKey(1), params)
cp = controller.reset(s_reset, params, controller.init_control_params, cr.PRNGKey(7))
from covo_mpc_amd.dynamics.dataclass import DeviceState
packed_d = torch.from_numpy(packed).to(device)
dref = s_reset.to_device(device)
key = cr.PRNGKey(1)
for i in range(45):
    key, k = cr.split(key)
    ds = DeviceState(packed=packed_d[i], pos_traj=dref.pos_traj, vel_traj=dref.vel_traj, time=int(host_states[i].time))
    u, cp, _ = controller(None, None, params, k, cp, {"noisy_state": ds})
torch.cuda.synchronize()
core = controller.core
T = core.time_phases
rows = [("whole step (graph)", T()), ("hessian (3 kernels)", T(2)),
        ("  jac", T(2, 1)), ("  chain + hyper-dual pairs + contraction", T(2, 2)), ("  gemm", T(2, 8)),
        ("sigma (all)", T(4)), ("  squarings with the Ritz evaluations inside", T(4, 15, 1)), ("  (+ scan launch: none on this path)", T(4, 15, 2)), ("  +newton-schulz", T(4, 15, 3)),
        ("sigma + gemm (product: eps drawn under finalize, tiled GEMM)", T(12)),
        ("noise gemm, in-kernel Philox (offline/MPPI path)", T(8)), ("rollout (+ softmax records)", T(16)),
        ("merge (softmax update)", T(32)), ("empty graph (floor)", T(0))]
for n, v in rows:
    print(f"{n:<62} {v:8.2f} us")
print("rollout reps=1", T(16, reps=1), "reps=5", T(16, reps=5), "reps=50", T(16, reps=50))
print("gemm+rollout", T(24), "gemm+rollout+softmax", T(56), "rollout+softmax", T(48))
pc = params.to_c()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): core.rollout(ds, pc, (0.0, 0.0, 0.0), False)
e0.record()
for _ in range(50): core.rollout(ds, pc, (0.0, 0.0, 0.0), False)
e1.record(); torch.cuda.synchronize()
print("eager b2b rollout", e0.elapsed_time(e1) / 50 * 1e3)

e0.record()
for i in range(50):
    key, k = cr.split(key)
    u, cp, _ = controller(None, None, params, k, cp, {"noisy_state": ds})
e1.record(); torch.cuda.synchronize()
print("real steps (same state), events:", e0.elapsed_time(e1) / 50 * 1e3, "us/step")
print("whole step via tool again", T())

from covo_mpc_amd import _lib
M = 128 * 128
names = ["SHIFT", "LMIN", "DELTA", "SCALE", "SUMLOGB", "ZBUF", "ITERS", "KWIN", "SQ", "SQ_DONE", "NS_DONE"]
for i in [5, 20, 44, 100, 200, 290]:
    ds_i = DeviceState(packed=packed_d[i % 60], pos_traj=dref.pos_traj, vel_traj=dref.vel_traj, time=int(host_states[i % 60].time))
    key, k = cr.split(key)
    u, cp, _ = controller(None, None, params, k, cp, {"noisy_state": ds_i})
    out_t = torch.zeros(24, dtype=torch.float64).pin_memory()
    _lib.check(core.lib.covo_debug_sigma_workspace(core.h, _lib.ptr(out_t), 11 * M, 24, core.stream()))
    torch.cuda.synchronize()
    o = out_t.numpy()
    t = o[16:21]
    print(i, "  ".join(f"{n}={o[j]:.5g}" for j, n in enumerate(names)),
          " finalize ticks: load %d chol %d logdet %d outputs %d" % tuple(t[k + 1] - t[k] for k in range(4)),
          " tail modes (2 = one XCD): squarings %d iterations %d" % (o[21], o[22]))
