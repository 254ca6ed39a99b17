#!/usr/bin/env python
"""Per-kernel micro-benchmark (events on the launch stream, back-to-back launches).
usage: python scripts/kbench.py [--N 8192 65536 ...] [--reps 50]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covo_mpc_amd as cm  # noqa: E402
from covo_mpc_amd import random as cr  # noqa: E402
from covo_mpc_amd.controllers._core import SamplingCore  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, nargs="+", default=[8192, 65536, 262144, 1048576])
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--lam", type=float, default=0.01)
    args = ap.parse_args()
    dev = "cuda:0"
    env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="gaussian",
                         disable_rollover_terminate=True, generate_noisy_state=True, device=dev)
    params = env.default_params
    obs, info, state = env.reset(cr.PRNGKey(1), params)
    ds = info["noisy_state"].to_device(dev)
    pc = params.to_c()
    rng = np.random.default_rng(0)
    A = rng.normal(size=(128, 128))
    R = torch.from_numpy(0.05 * (A + A.T)).to(dev).reshape(1, 128, 128)
    print(f"{'N':>8} {'kernel':<16} {'us':>9}  note")
    for N in args.N:
        core = SamplingCore(N, 32, args.lam, 1.0, device=dev, trust_clipped=True)
        am = torch.tensor([-0.3378, 0, 0, 0], device=dev).repeat(32)
        Sigma, L = core.sigma(R, 0.5)
        L = L[0].contiguous()
        t = timeit(lambda: core.randn((1, 2)), args.reps)
        print(f"{N:>8} {'randn':<16} {t:9.2f}  {N*512/t/1e3:.0f} GB/s written")
        t = timeit(lambda: core.noise_gemm(L, am), args.reps)
        print(f"{N:>8} {'noise_gemm':<16} {t:9.2f}  {N*32768/t/1e6:.1f} TFLOP/s dense-equiv, {N*1024/t/1e3:.0f} GB/s")
        t = timeit(lambda: core.noise_gemm_philox(L, am, (1, 2)), args.reps)
        print(f"{N:>8} {'noise_gemm+rng':<16} {t:9.2f}  {N*32768/t/1e6:.1f} TFLOP/s dense-equiv (epsilon drawn in-kernel)")
        t = timeit(lambda: core.rollout(ds, pc, (0.0, 0.0, 0.0), False), args.reps)
        print(f"{N:>8} {'rollout':<16} {t:9.2f}  {N*516/t/1e3:.0f} GB/s = {N*516/t/1e3/8000*100:.1f}% of 8 TB/s")
        t = timeit(lambda: core.rollout(ds, pc, (0.0, 0.0, 0.0), True), args.reps)
        print(f"{N:>8} {'rollout+stats':<16} {t:9.2f}")
        t = timeit(lambda: core.update(am, 1.0), args.reps)
        print(f"{N:>8} {'softmax_update':<16} {t:9.2f}  (lam={args.lam}: zero-weight groups skipped)")
        if N == args.N[0]:
            t = timeit(lambda: core.hessian(ds.packed, ds, pc, am), args.reps)
            print(f"{'-':>8} {'hessian':<16} {t:9.2f}")
            t = timeit(lambda: core.sigma(R, 0.5), max(args.reps // 5, 3))
            print(f"{'-':>8} {'sigma (eigh-free)':<16} {t:9.2f}")
            t = timeit(lambda: core.sigma(R, 0.5, method="jacobi"), max(args.reps // 5, 3))
            print(f"{'-':>8} {'sigma (jacobi)':<16} {t:9.2f}")
            t = timeit(lambda: core.shift_mean(am), args.reps)
            print(f"{'-':>8} {'shift_mean':<16} {t:9.2f}")
        del core


if __name__ == "__main__":
    main()
