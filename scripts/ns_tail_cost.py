#!/usr/bin/env python
"""Cost of a live Newton-Schulz iteration inside the tail launch (two grid barriers) against its two separate launches:
covo_sigma on a closed-loop-like matrix (10 iterations), graph-free eager timing, tail = 0 / 2 / 11 iterations."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covo_mpc_amd import _lib
from covo_mpc_amd.controllers._core import SamplingCore
core = SamplingCore(256, 32, 0.01, 1.0, device="cuda:0")
lib = core.lib
rng = np.random.default_rng(5)
n = 128
Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
w = np.concatenate([[-2.0, -1.1, -0.7], np.geomspace(0.01, 900.0, n - 3)])
R = torch.from_numpy(np.ascontiguousarray((Q * w) @ Q.T)).cuda()[None]
for tail in ((0, 0), (0, 2), (0, 64), (64, 0), (64, 64)):
    _lib.check(lib.covo_debug_set_ns_tail(core.h, *tail))
    for _ in range(5): core.sigma(R, 0.5)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        core.sigma(R, 0.5)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10): core.sigma(R, 0.5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    out_t = torch.zeros(24, dtype=torch.float64).pin_memory()
    _lib.check(lib.covo_debug_sigma_workspace(core.h, _lib.ptr(out_t), 11 * n * n, 24, core.stream()))
    torch.cuda.synchronize()
    print(f"tail (squarings, iterations) = {tail}: covo_sigma {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us in a graph  (squarings {int(out_t[7])}, NS iterations {int(out_t[6])}; tail modes {int(out_t[21])} / {int(out_t[22])} [2 = one XCD])")
    if os.environ.get("NS_STAMPS") and tail[1]:  # library built with -DNS_STAMPS: the seams of workgroup 0's phases, 10 ns ticks
        SC_STAMPS = int(os.environ["NS_STAMPS"])
        st = torch.zeros(192, dtype=torch.float64).pin_memory()
        _lib.check(lib.covo_debug_sigma_workspace(core.h, _lib.ptr(st), 11 * n * n + SC_STAMPS, 192, core.stream()))
        torch.cuda.synchronize()
        v = [x for x in st.tolist() if x >= 0]
        print("   stamps (us):", " ".join(f"{x / 100.0:.2f}" for x in v))
        print("   deltas (us):", " ".join(f"{(b - a) / 100.0:.2f}" for a, b in zip(v, v[1:])))
_lib.check(lib.covo_debug_set_ns_tail(core.h, -1, -1))
