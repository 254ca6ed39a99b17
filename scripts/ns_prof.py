#!/usr/bin/env python
"""Eigh-free Sigma pipeline: iteration counts, scalars and the finalize kernel's clock stamps
(s_memtime, 100 MHz -> 10 ns/tick), plus event timing of covo_sigma launched eagerly."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covo_mpc_amd import _lib
from covo_mpc_amd.controllers._core import SamplingCore
core = SamplingCore(256, 32, 0.01, 1.0, device="cuda:0")
rng = np.random.default_rng(0)
n = 128
A = rng.normal(size=(n, n))
mats = {"generic": 0.05 * (A + A.T)}
G = rng.standard_normal((n, n)); Q, _ = np.linalg.qr(G)
w = np.concatenate([np.abs(rng.standard_normal(20)) * 50, rng.standard_normal(108) * 0.05])
mats["covo-like"] = (Q * w) @ Q.T
M = n * n
SC0 = 11 * M
names = ["SHIFT", "LMIN", "DELTA", "SCALE", "LOGDET", "ZBUF", "ITERS", "XBUF", "SQ", "SQ_DONE", "NS_DONE", "FRO2", "TRACE", "GERSH", "N0"]
for name, Rm in mats.items():
    R = torch.from_numpy(Rm).cuda()
    Sig = torch.empty(n, n, device="cuda"); L = torch.empty(n, n, device="cuda")
    def run():
        _lib.check(core.lib.covo_sigma(core.h, _lib.ptr(R), 1, 0.5, _lib.ptr(Sig), _lib.ptr(L), core.stream()))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    out_t = torch.zeros(64, dtype=torch.float64).pin_memory()
    _lib.check(core.lib.covo_debug_sigma_workspace(core.h, _lib.ptr(out_t), SC0, 64, core.stream()))
    torch.cuda.synchronize()
    out = out_t.numpy()
    print(f"== {name}: covo_sigma eager {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
    print("   " + "  ".join(f"{k}={out[i]:.6g}" for i, k in enumerate(names)))
    t = out[16:21]
    lab = ["load Z", "chol(Z)", "logdet+scalars", "outputs"]
    print("   finalize (s_memtime ticks ~ core cycles): " + "  ".join(f"{lab[i]} {t[i + 1] - t[i]:.0f}" for i in range(4)))
    w_ = np.linalg.eigvalsh(0.5 * (Rm + Rm.T))
    print(f"   lmin err {out[1] - w_[0]:.3e}")
