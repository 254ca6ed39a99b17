"""covo_sigma on a few closed-loop-like Hessians, for `rocprofv3 --kernel-trace --stats` (per-kernel durations of the chain)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from covo_mpc_amd.controllers._core import SamplingCore

core = SamplingCore(N=8192, H=32, lam=0.01, discount=1.0, device="cuda")
rng = np.random.default_rng(0)
Q, _ = np.linalg.qr(rng.standard_normal((128, 128)))
lam = np.concatenate([[-3.6, -3.1, -2.0], rng.uniform(-1.0, 12.0, 125)])
R = torch.from_numpy((Q * lam) @ Q.T).to("cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    Sigma, L = core.sigma(R[None], 0.5)
torch.cuda.synchronize()
print("ok", float(Sigma.abs().max()))
