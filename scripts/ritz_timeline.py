#!/usr/bin/env python
"""Timeline of the squaring launch with the Rayleigh-Ritz evaluations inside (library built with -DNS_EVAL_STAMPS:
`make -C covo_mpc_amd/csrc VARIANT=evst HIPFLAGS+=-DNS_EVAL_STAMPS`, COVO_HIP_LIB=...): when X_k is complete, when its
evaluation starts / ends / is decided, when the chain leaves.  Real Hessians from tests/golden/hessians_r03.npz."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covo_mpc_amd import _lib
from covo_mpc_amd.controllers._core import SamplingCore
core = SamplingCore(256, 32, 0.01, 1.0, device="cuda:0")
lib = core.lib
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hessians_r03.npz"))
mats = [m for k in g.files for m in g[k]]
SC_STAMPS = 3520
for i in (0, 5, 11):
    R = torch.from_numpy(mats[i][None].copy()).cuda()
    for _ in range(3): core.sigma(R, 0.5)
    torch.cuda.synchronize()
    st = torch.zeros(96, dtype=torch.float64).pin_memory()
    sc = torch.zeros(32, dtype=torch.float64).pin_memory()
    _lib.check(lib.covo_debug_sigma_workspace(core.h, _lib.ptr(st), 11 * 128 * 128 + SC_STAMPS + 96, 96, core.stream()))
    _lib.check(lib.covo_debug_sigma_workspace(core.h, _lib.ptr(sc), 11 * 128 * 128, 32, core.stream()))
    torch.cuda.synchronize()
    v = st.numpy(); t0 = v[0]
    us = lambda x: (x - t0) / 100.0
    print(f"matrix {i}: kwin {int(sc[7])}, squarings started {int(sc[8])}, NS iterations {int(sc[6])}; chain workgroup 0 leaves at {us(v[1]):.2f} us")
    print("   X_k complete (chain wg 0 past the barrier):", " ".join(f"{k}:{us(v[48 + k]):.1f}" for k in range(2, 17) if v[48 + k] >= t0))
    for k in range(2, 17):
        a, b, c = v[2 + 3 * (k - 2): 5 + 3 * (k - 2)]
        if a >= t0:
            print(f"   k={k:2d}: seen {us(a):6.2f}  evaluated {us(b) if b >= a else float('nan'):6.2f}  decided {us(c) if c >= a else float('nan'):6.2f}")
    print("   inside the evaluation of X_8 (us from its start): " + " ".join(f"{(v[70 + j] - v[70]) / 100.0:.2f}" for j in range(5)) +
          f"  [picks+MGS | A.V | H | Jacobi];  evaluated at +{(v[3 + 3 * 6] - v[70]) / 100.0:.2f}")
