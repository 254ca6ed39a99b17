import sys, time, torch
sys.path.insert(0, '/root/repo')
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr
env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="gaussian", disable_rollover_terminate=True,
                     generate_noisy_state=True, device="cuda:0")
c, cp = cm.envs.get_controller(env, "covo-offline", "N8192_H32_lam0.01", device="cuda:0")
params = env.default_params
obs, info, state = env.reset(cr.PRNGKey(1), params)
for i in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    cp = c.reset(state, params, c.init_control_params, cr.PRNGKey(2))
    torch.cuda.synchronize(); print("covo-offline reset (300-step Sigma table): %.1f ms" % ((time.perf_counter() - t) * 1e3))
