import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr, _lib
from oracle import ref_np as R
env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="gaussian", disable_rollover_terminate=True, generate_noisy_state=True, device="cuda:0")
controller, cp = cm.envs.get_controller(env, "covo-online", "N2048_H32_lam0.01", device="cuda:0")
params = env.default_params
obs, info, state = env.reset(cr.PRNGKey(1), params)
key = cr.PRNGKey(3); core = controller.core
for step in range(4):
    key, k_act, k_step = cr.split(key, 3)
    u, cp, cinfo = controller(obs, state, params, k_act, cp, info)
    obs, state, reward, done, info = env.step(k_step, state, u.cpu().numpy(), params)
ds = info["noisy_state"].to_device("cuda:0")
am = core.shift_mean(cp.a_mean.reshape(-1))
Rg = core.hessian(ds.packed, ds, params.to_c(), am)
Rm = Rg[0].cpu().numpy()
ref = R.optimize_sigma(Rm, 0.5, 32, 4)
w = np.linalg.eigvalsh((Rm+Rm.T)/2); print("eig", w[:4], w[-3:])
for m in ("ns", "jacobi"):
    S, L = core.sigma(Rg, 0.5, method=m); S = S[0].cpu().numpy()
    print(m, "rel err", np.linalg.norm(S-ref)/np.linalg.norm(ref))
M = 128*128
def grab(off, cnt):
    out = torch.empty(cnt, dtype=torch.float64, device="cuda")
    _lib.check(core.lib.covo_debug_sigma_workspace(core.h, _lib.ptr(out), off, cnt, core.stream()))
    torch.cuda.synchronize(); return out.cpu().numpy()
S, L = core.sigma(Rg, 0.5, method="ns")
sc = grab(12*M, 64)
print("shift", sc[24], "lmin", sc[25], "err", sc[25]-w[0], "scale", sc[27], "logdet", sc[28], "iters", sc[30])
Bm = (Rm+Rm.T)/2 + (0.01 - w[0])*np.eye(128)
print("true logdet", np.linalg.slogdet(Bm)[1], "cond", np.linalg.cond(Bm))
print("normsq", sc[:13]); print("err2", sc[32:56])
Z = grab((9 if sc[29] else 8)*M, M).reshape(128,128)
wb, U = np.linalg.eigh(Bm); Zref = (U/np.sqrt(wb/sc[27]))@U.T
print("Z abs err", np.abs(Z-Zref).max(), "Z max", np.abs(Zref).max(), "rel fro", np.linalg.norm(Z-Zref)/np.linalg.norm(Zref))
# error decomposition in eigenbasis
D = U.T@(Z-Zref)@U
print("err in eigenbasis: diag max", np.abs(np.diag(D)).max(), "at", np.argmax(np.abs(np.diag(D))), "offdiag max", np.abs(D-np.diag(np.diag(D))).max())
print("diag err first 5", np.diag(D)[:5], "Zref eig first", (1/np.sqrt(wb/sc[27]))[:3])
