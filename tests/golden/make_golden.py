"""Regenerates tests/golden/rollout_small.npz from the fp64 numpy oracle (oracle/ref_np.py).

The reference itself cannot run in the build container (no jax; PARITY UNPINNED), so these vectors
pin the HIP kernels and the C oracle to the line-by-line restatement -- inputs and expected
outputs only.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_np as R  # noqa: E402
from tests.conftest import make_problem  # noqa: E402


def main():
    out = {}
    for name, kw in {"mid": dict(seed=0, time=37), "late": dict(seed=1, time=285), "start": dict(seed=2, time=0)}.items():
        s, p, rng = make_problem(**kw)
        N, H = 64, 32
        a = np.clip(R.hover_action(p, H, np.float64)[None] + 0.5 * rng.normal(size=(N, H, 4)), -1, 1)
        a = a.astype(np.float32).astype(np.float64)
        if name == "late":
            a[:8, :, 1] = 1.0  # spin some samples; others leave the |pos|<3 box below
        f_shared = np.array([0.01, -0.02, 0.03], dtype=np.float32).astype(np.float64) if name == "mid" else np.zeros(3)
        disc = 0.97 if name == "mid" else 1.0
        if name == "start":
            s = s.replace(pos=s.pos + np.array([2.9, 0, 0]), vel=s.vel + np.array([3.0, 0, 0]))  # forces |pos|>3 freeze
        cost, rewards, poses = R.rollout(s, p, a, disc, f_shared)
        out[f"{name}_state"] = np.concatenate([s.pos, s.vel, s.quat, s.omega, s.f_disturb, s.pos_tar, s.vel_tar])
        out[f"{name}_time"] = np.int32(s.time)
        out[f"{name}_pos_traj"], out[f"{name}_vel_traj"] = s.pos_traj, s.vel_traj
        out[f"{name}_a"], out[f"{name}_f_shared"], out[f"{name}_discount"] = a, f_shared, np.float64(disc)
        out[f"{name}_cost"], out[f"{name}_rewards"], out[f"{name}_poses"] = cost, rewards, poses
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rollout_small.npz"), **out)
    print({k: v.shape for k, v in out.items() if k.endswith("cost")})


if __name__ == "__main__":
    main()
