import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr, _lib
from oracle import ref_np as R
env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="gaussian", disable_rollover_terminate=True, generate_noisy_state=True, device="cuda:0")
controller, cp = cm.envs.get_controller(env, "covo-online", "N2048_H32_lam0.01", device="cuda:0")
params = env.default_params
obs, info, state = env.reset(cr.PRNGKey(1), params)
cp = controller.reset(state, params, controller.init_control_params, cr.PRNGKey(2))
key = cr.PRNGKey(3); core = controller.core
M = 128*128
def grab(off, cnt):
    out = torch.empty(cnt, dtype=torch.float64, device="cuda")
    _lib.check(core.lib.covo_debug_sigma_workspace(core.h, _lib.ptr(out), off, cnt, core.stream()))
    torch.cuda.synchronize(); return out.cpu().numpy()
for step in range(4):
    key, k_act, k_step = cr.split(key, 3)
    ds = info["noisy_state"].to_device("cuda:0")
    am = core.shift_mean(cp.a_mean.reshape(-1))
    Rg = core.hessian(ds.packed, ds, params.to_c(), am); Rm = Rg[0].cpu().numpy()
    u, cp, cinfo = controller(obs, state, params, k_act, cp, info)
    S = cp.a_cov.cpu().numpy().astype(np.float64)
    sc = grab(12*M, 64)
    w = np.linalg.eigvalsh((Rm+Rm.T)/2)
    ws = np.linalg.eigvalsh((S+S.T)/2)
    print(f"step {step}: R finite {np.isfinite(Rm).all()} |R|max {np.abs(Rm).max():.3g} eig [{w[0]:.4g},{w[1]:.4g}..{w[-1]:.4g}]  Sigma finite {np.isfinite(S).all()} eig [{ws[0]:.3g}..{ws[-1]:.3g}]")
    print(f"   lmin {sc[25]:.6g} (err {sc[25]-w[0]:.2e}) scale {sc[27]:.4g} logdet {sc[28]:.6g} iters {sc[30]} err2 tail {sc[32+int(sc[30])-2:32+int(sc[30])+1]}")
    if np.isfinite(Rm).all():
        ref = R.optimize_sigma(Rm, 0.5, 32, 4); print("   rel err vs oracle", np.linalg.norm(S-ref)/np.linalg.norm(ref))
    obs, state, reward, done, info = env.step(k_step, state, u.cpu().numpy(), params)
