import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr
env = cm.envs.Quad3D(task="tracking_zigzag", obs_type="quad", enable_randomizer=False, disturb_type="gaussian",
                     disable_rollover_terminate=True, generate_noisy_state=True, device="cuda:0")
params = env.default_params
controller, cp = cm.envs.get_controller(env, "covo-offline", "N65536_H32_lam0.01", device="cuda:0", compute_info=False)
controller.alias_outputs = True
core = controller.core
ep = cm.envs.DeviceEpisode(env, cr.PRNGKey(21), params, (core.lib, core.h), core.device)
cp = controller.reset(ep.state0, params, controller.init_control_params, cr.PRNGKey(22))
key = cr.PRNGKey(23)
torch.cuda.synchronize()
tc = te = 0.0
t0 = time.perf_counter()
for i in range(300):
    key, k_act, k_step = cr.split(key, 3)
    a = time.perf_counter()
    u, cp, _ = controller(None, None, params, k_act, cp, {"noisy_state": ep.noisy_state})
    b = time.perf_counter()
    ep.step(k_step, u)
    c = time.perf_counter()
    tc += b - a; te += c - b
torch.cuda.synchronize()
print("total/step us", (time.perf_counter() - t0) / 300 * 1e6, "controller host us", tc / 300 * 1e6, "env host us", te / 300 * 1e6)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for i in range(100):
    key, k_act, k_step = cr.split(key, 3)
    u, cp, _ = controller(None, None, params, k_act, cp, {"noisy_state": ep.noisy_state})
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
def timeit(f, n=100):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
def ctrl():
    global key, cp
    key, k_act, k_step = cr.split(key, 3)
    u, cp, _ = controller(None, None, params, k_act, cp, {"noisy_state": ep.noisy_state})
print("controller only (time=None):", timeit(ctrl))
ustat = torch.zeros(4, device="cuda")
print("env step only:", timeit(lambda: (setattr(ep, "n_steps", 0), ep.step(cr.PRNGKey(3), ustat))))
tab = cp.a_cov_offline
def gather():
    t_idx = ep.noisy[25:26].view(torch.int32).long().clamp(0, tab.shape[0] - 1)
    return tab.index_select(0, t_idx)[0]
print("gather only:", timeit(gather))
def both_dummy():
    ctrl(); ustat.add_(1.0)
print("controller + dummy torch kernel:", timeit(both_dummy))
def both_env():
    global key, cp
    key, k_act, k_step = cr.split(key, 3)
    u, cp, _ = controller(None, None, params, k_act, cp, {"noisy_state": ep.noisy_state})
    ep.n_steps = 0
    ep.step(k_step, ustat)
print("controller + env(static action):", timeit(both_env))
def both_env_u():
    global key, cp
    key, k_act, k_step = cr.split(key, 3)
    u, cp, _ = controller(None, None, params, k_act, cp, {"noisy_state": ep.noisy_state})
    ep.n_steps = 0
    ep.step(k_step, u)
print("controller + env(u):", timeit(both_env_u))
# fresh episode, 150 closed-loop steps, then phase times at that state
ep = cm.envs.DeviceEpisode(env, cr.PRNGKey(21), params, (core.lib, core.h), core.device)
cp = controller.reset(ep.state0, params, controller.init_control_params, cr.PRNGKey(22))
key = cr.PRNGKey(23)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(150):
    key, k_act, k_step = cr.split(key, 3)
    u, cp, _ = controller(None, None, params, k_act, cp, {"noisy_state": ep.noisy_state})
    ep.step(k_step, u)
torch.cuda.synchronize(); print("150 closed-loop steps: us/step", (time.perf_counter() - t0) / 150 * 1e6)
T_ = core.time_phases
print("phases: all", T_(), "begin", T_(1), "gemm", T_(8), "rollout", T_(16), "softmax", T_(32))
c = core.cost.cpu().numpy()
print("cost min", c.min(), "n within 0.87 of min:", int((c - c.min() < 0.87).sum()), "n live groups of 8:", int(((c.reshape(-1, 8) - c.min() < 0.87).any(1)).sum()))
import bench
r = bench.closed_loop(env, controller, params, 300)
print(r)
ep = cm.envs.DeviceEpisode(env, cr.PRNGKey(21), params, (core.lib, core.h), core.device)
cp = controller.reset(ep.state0, params, controller.init_control_params, cr.PRNGKey(22))
key = cr.PRNGKey(23)
for blk in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(50):
        key, k_act, k_step = cr.split(key, 3)
        u, cp, _ = controller(None, None, params, k_act, cp, {"noisy_state": ep.noisy_state})
        ep.step(k_step, u)
    torch.cuda.synchronize(); print("steps", blk * 50, "us/step", (time.perf_counter() - t0) / 50 * 1e6, "softmax", core.time_phases(32), "rollout", core.time_phases(16))
