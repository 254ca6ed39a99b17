#!/usr/bin/env python
"""Clock stamps of the adjoint-Hessian chain kernel (s_memtime ~ core clock)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from covo_mpc_amd import _lib
from covo_mpc_amd.controllers._core import SamplingCore
from covo_mpc_amd.dynamics.dataclass import EnvParams3D
from conftest import make_problem
from test_gpu_parity import dev_state
from oracle import ref_np as R
s, p, rng = make_problem(seed=0, time=37)
a = (R.hover_action(p, 32, np.float64) + 0.1 * rng.normal(size=(32, 4))).astype(np.float32)
core = SamplingCore(256, 32, 0.01, 1.0, device="cuda:0")
ds = dev_state(s)
am = torch.from_numpy(a.reshape(-1)).cuda()
for _ in range(3):
    Rm = core.hessian(ds.packed, ds, EnvParams3D().to_c(), am)
torch.cuda.synchronize()
WS_LAM = 32 * 16 + 32 * 13 * 17 + 32 * 16
out = torch.zeros(33 * 16, dtype=torch.float64).pin_memory()
_lib.check(core.lib.covo_debug_hess_workspace(core.h, _lib.ptr(out), WS_LAM, 33 * 16, core.stream()))
torch.cuda.synchronize()
o = out.numpy().reshape(33, 16)
print("wave: after-fill / end (ticks)")
for w in range(9):
    print(w, int(o[w, 13]), int(o[w, 14]))
