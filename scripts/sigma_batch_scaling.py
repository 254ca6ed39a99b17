#!/usr/bin/env python
"""covo_sigma on `batch` copies (scaled) of one closed-loop-like matrix, eager: run under rocprofv3 --kernel-trace --stats to see
how the chain's launches scale with the batch (4 matrices share an XCD at 32).  usage: sigma_batch_scaling.py <batch>"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covo_mpc_amd.controllers._core import SamplingCore
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
core = SamplingCore(256, 32, 0.01, 1.0, device="cuda:0")
rng = np.random.default_rng(5)
n = 128
Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
w = np.concatenate([[-2.0, -1.1, -0.7], np.geomspace(0.01, 900.0, n - 3)])
R1 = (Q * w) @ Q.T
R = torch.from_numpy(np.ascontiguousarray(np.stack([R1 * (1 + 0.01 * i) for i in range(batch)]))).cuda()
for _ in range(3): core.sigma(R, 0.5, batch=batch)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): core.sigma(R, 0.5, batch=batch)
e1.record(); torch.cuda.synchronize()
print(f"batch {batch:3d}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per covo_sigma (eager)")
