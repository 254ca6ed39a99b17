#!/usr/bin/env python
"""Host-side cost of one controller __call__ (cProfile over teacher-forced steps): what bounds the small configs."""
import cProfile, pstats, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr
from covo_mpc_amd.dynamics.dataclass import DeviceState
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "mppi"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = "cuda:0"
env = cm.envs.Quad3D(task="tracking_zigzag", obs_type="quad", enable_randomizer=False, disturb_type="gaussian",
                     disable_rollover_terminate=True, generate_noisy_state=True, device=dev)
params = env.default_params
controller, cp = cm.envs.get_controller(env, name, f"N{N}_H32_lam0.01", device=dev, compute_info=False)
state0, packed, host_states = bench.make_states(env, params, 300, seed=1)
obs0, info0, s_reset = env.reset(cr.PRNGKey(1), params)
cp = controller.reset(s_reset, params, controller.init_control_params, cr.PRNGKey(7))
packed_d = torch.from_numpy(packed).to(dev)
dref = s_reset.to_device(dev)
ds = [DeviceState(packed=packed_d[i], pos_traj=dref.pos_traj, vel_traj=dref.vel_traj, time=int(host_states[i].time)) for i in range(300)]
controller.alias_outputs = True
key = cr.PRNGKey(1)
def run(n, key, cp):
    for i in range(n):
        key, k = cr.split(key)
        u, cp, _ = controller(None, None, params, k, cp, {"noisy_state": ds[i % 300]})
    return key, cp
key, cp = run(50, key, cp)
torch.cuda.synchronize()
t0 = time.perf_counter(); key, cp = run(2000, key, cp); th = time.perf_counter() - t0
torch.cuda.synchronize(); tt = time.perf_counter() - t0
print(f"{name} N={N}: host enqueue {1e6 * th / 2000:.1f} us/step, wall {1e6 * tt / 2000:.1f} us/step")
pr = cProfile.Profile(); pr.enable(); key, cp = run(2000, key, cp); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
