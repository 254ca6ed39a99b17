import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr
dev = "cuda:0"
env = cm.envs.Quad3D(task="tracking", obs_type="quad_params", enable_randomizer=True, disturb_type="gaussian",
                     disable_rollover_terminate=True, generate_noisy_state=True, device=dev)
for E, N in ((32, 4096), (16, 8192), (8, 16384), (4, 32768), (2, 65536), (64, 2048)):
    c0, _ = cm.envs.get_controller(env, "covo-online", f"N{N}_H32_lam0.01", device=dev, compute_info=False)
    cp0 = c0.init_control_params
    c0.core.close()
    params = [env.sample_params(cr.PRNGKey(1000 + g)) for g in range(E)]
    b = cm.controllers.BatchedCoVOController(env, E, N, 32, 0.01, discount=cp0.discount, gamma_mean=cp0.gamma_mean,
                                             sample_sigma=cp0.sample_sigma, a_mean_init=cp0.a_mean, device=dev)
    ep = cm.envs.BatchedDeviceEpisode(env, [cr.PRNGKey(2000 + g) for g in range(E)], params, (b.core.lib, b.core.h), dev)
    rngs = np.stack([np.asarray(cr.PRNGKey(3000 + g)) for g in range(E)])
    rngs = b.run_episode(ep, rngs, 10)
    torch.cuda.synchronize()
    print(f"E={E} N={N}: rollout {min(b.time_phases(16) for _ in range(3)):.2f} us  gemm {min(b.time_phases(8) for _ in range(3)):.2f} us", flush=True)
    b.core.close()
