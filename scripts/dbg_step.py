import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr
name = sys.argv[1]
env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="gaussian", disable_rollover_terminate=True, generate_noisy_state=True, device="cuda:0")
controller, cp = cm.envs.get_controller(env, name, "N4096_H32_lam0.01", device="cuda:0", compute_info=False)
params = env.default_params
obs, info, state = env.reset(cr.PRNGKey(1), params)
cp = controller.reset(state, params, controller.init_control_params, cr.PRNGKey(2))
torch.cuda.synchronize(); print("reset ok")
key = cr.PRNGKey(3)
for step in range(6):
    key, k_act, k_step = cr.split(key, 3)
    u, cp, cinfo = controller(obs, state, params, k_act, cp, info)
    torch.cuda.synchronize(); print("step", step, u.cpu().numpy())
    obs, state, reward, done, info = env.step(k_step, state, u.cpu().numpy(), params)
