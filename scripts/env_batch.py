#!/usr/bin/env python
"""BASELINE configs[4] on one GPU: E independent env instances (lissajous `tracking`, domain-randomised parameters), each
its own covo-online MPC problem with N samples ("replicas only"): one controller handle + graph per
instance, instances spread over S HIP streams so their latency-bound Sigma chains overlap.  Prints env-steps/s."""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import covo_mpc_amd as cm
from covo_mpc_amd import random as cr

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=32)
ap.add_argument("--streams", type=int, default=8)
ap.add_argument("--N", type=int, default=4096)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--controller", default="covo-online")
args = ap.parse_args()
dev = "cuda:0"
env = cm.envs.Quad3D(task="tracking", obs_type="quad_params", enable_randomizer=True, disturb_type="gaussian",
                     disable_rollover_terminate=True, generate_noisy_state=True, device=dev)
streams = [torch.cuda.Stream() for _ in range(args.streams)]
inst = []
for e in range(args.envs):
    params = env.sample_params(cr.PRNGKey(100 + e))
    with torch.cuda.stream(streams[e % args.streams]):
        c, _ = cm.envs.get_controller(env, args.controller, f"N{args.N}_H32_lam0.01", device=dev, compute_info=False)
        c.alias_outputs = True
        ep = cm.envs.DeviceEpisode(env, cr.PRNGKey(200 + e), params, (c.core.lib, c.core.h), dev)
        cp = c.reset(ep.state0, params, c.init_control_params, cr.PRNGKey(2))
        cp, rng = c.run_episode(ep, params, cp, cr.PRNGKey(300 + e), 3)  # eager, capture, replay
    inst.append([c, ep, params, cp, rng])
torch.cuda.synchronize()
import threading
def worker(sid):
    # one host thread per stream: covo_run_episode is a single ctypes call (the GIL is released while it enqueues)
    with torch.cuda.stream(streams[sid]):
        for e in range(sid, args.envs, args.streams):
            c, ep, params, cp, rng = inst[e]
            inst[e][3], inst[e][4] = c.run_episode(ep, params, cp, rng, args.steps)
t0 = time.perf_counter()
threads = [threading.Thread(target=worker, args=(sid,)) for sid in range(args.streams)]
for th in threads: th.start()
for th in threads: th.join()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
el = time.perf_counter() - t0
errs = [it[1].read_log()[3:, 1].mean() for it in inst]
print(f"{args.controller} E={args.envs} N={args.N} streams={args.streams}: {args.envs * args.steps / el:.0f} env-steps/s "
      f"({1e6 * el / args.steps:.0f} us per batched step; host enqueue {1e3 * t_host:.0f} ms of {1e3 * el:.0f} ms); "
      f"mean err_pos {np.mean(errs):.3f} m")
