#!/usr/bin/env python
"""covo_sigma on one real Hessian, 200 calls (for rocprofv3 --kernel-trace --stats A/Bs of the chain's kernels)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covo_mpc_amd.controllers._core import SamplingCore
core = SamplingCore(256, 32, 0.01, 1.0, device="cuda:0")
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hessians_r03.npz"))
mats = [m for k in g.files for m in g[k]]
R = torch.from_numpy(mats[int(sys.argv[1]) if len(sys.argv) > 1 else 0][None].copy()).cuda()
for _ in range(200): core.sigma(R, 0.5)
torch.cuda.synchronize()
