#!/usr/bin/env python
"""Phase breakdown of sigma_kernel (s_memtime ticks; 100 MHz constant clock on gfx950 -> 10 ns/tick)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covo_mpc_amd import _lib
from covo_mpc_amd.controllers._core import SamplingCore
core = SamplingCore(256, 32, 0.01, 1.0, device="cuda:0")
rng = np.random.default_rng(0)
A = rng.normal(size=(128, 128)); R = torch.from_numpy(0.05 * (A + A.T)).cuda()
Sig = torch.empty(128, 128, device="cuda"); L = torch.empty(128, 128, device="cuda")
ticks = torch.zeros(32, dtype=torch.int64, device="cuda")
for _ in range(3):
    _lib.check(core.lib.covo_sigma_profile(core.h, _lib.ptr(R), 0.5, _lib.ptr(Sig), _lib.ptr(L), _lib.ptr(ticks), core.stream()))
torch.cuda.synchronize()
t = ticks.cpu().numpy()
ns = int(t[24]); t0 = t[0]
print("sweeps:", ns)
print("load+shift   %8d ticks" % (t[1] - t0))
prev = t[1]
for i in range(ns):
    print("sweep %2d     %8d ticks" % (i, t[2 + i] - prev)); prev = t[2 + i]
print("spectrum map %8d ticks" % (t[21] - t[20]))
print("H H^T        %8d ticks" % (t[22] - t[21]))
print("cholesky     %8d ticks" % (t[23] - t[22]))
print("total        %8d ticks" % (t[23] - t0))
