import sys; sys.path.insert(0, "/root/repo")
import torch, numpy as np
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr
from covo_mpc_amd.dynamics.dataclass import DeviceState
import bench
for name, N in (("covo-offline", 8192), ("mppi", 1024), ("covo-offline", 65536)):
    dev = "cuda:0"
    env = cm.envs.Quad3D(task="tracking_zigzag", obs_type="quad", enable_randomizer=False, disturb_type="gaussian",
                         disable_rollover_terminate=True, generate_noisy_state=True, device=dev)
    params = env.default_params
    c, cp = cm.envs.get_controller(env, name, f"N{N}_H32_lam0.01", device=dev, compute_info=False)
    state0, packed, host_states = bench.make_states(env, params, 60, seed=1)
    obs0, info0, s_reset = env.reset(cr.PRNGKey(1), params)
    cp = c.reset(s_reset, params, c.init_control_params, cr.PRNGKey(7))
    packed_d = torch.from_numpy(packed).to(dev); dref = s_reset.to_device(dev)
    key = cr.PRNGKey(1)
    for i in range(10):
        key, k = cr.split(key)
        ds = DeviceState(packed=packed_d[i], pos_traj=dref.pos_traj, vel_traj=dref.vel_traj, time=int(host_states[i].time))
        u, cp, _ = c(None, None, params, k, cp, {"noisy_state": ds})
    torch.cuda.synchronize()
    T = c.core.time_phases
    print(f"{name} N={N}: whole graph {T():.2f} us; noise {T(8):.2f}; rollout(+records) {T(16):.2f}; merge {T(32):.2f}; empty {T(0):.2f}")
