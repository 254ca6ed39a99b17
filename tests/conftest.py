import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def make_problem(seed=0, time=37, dtype=np.float64, task="zigzag"):
    """A deterministic mid-episode tracking state + default params (oracle types)."""
    from oracle import ref_np as R
    rng = np.random.default_rng(seed)
    p = R.Params().fp32()
    if task == "zigzag":
        pos, vel, acc = R.generate_zigzag_traj(300, p.dt, rng)
    elif task == "lissa":
        pos, vel, acc = R.generate_lissa_traj(300, p.dt, rng)
    else:
        pos, vel, acc = R.generate_fixed_traj(300, p.dt)
    s = R.zero_state(pos, vel, acc, rng.uniform(-0.2, 0.2, 3), dtype=np.float64)
    t = min(time, pos.shape[0] - 1)
    q = np.array([0.05, -0.03, 0.02, 0.99]) + 0.001 * rng.normal(size=4)  # noisy, un-normalised (covo.py:198)
    s = s.replace(time=time, pos=pos[t] + 0.05 * rng.normal(size=3), vel=vel[t] * 0.9 + 0.05 * rng.normal(size=3),
                  quat=q, omega=0.3 * rng.normal(size=3), pos_tar=pos[t], vel_tar=vel[t], acc_tar=acc[t])
    # the reference is fp32: every input is an fp32-representable number
    s = s.astype(np.float32).astype(dtype)
    return s, p, rng


@pytest.fixture
def problem():
    return make_problem()
