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


def table_ref(p, s, kind, draws, H=32):
    """The table of csrc/disturb.hip from the oracle's model functions: row k = {g_k, c_k},
    f_k = c_drag drag(vel_{k-1}) + c_k f_{k-1} + g_k."""
    from oracle import ref_np as R
    tab = np.zeros((H, 4))
    f = np.asarray(s.f_disturb, dtype=np.float64)
    for k in range(H - 1):
        sk = s.replace(time=s.time + k, f_disturb=f)
        hit = (s.time + k) % p.disturb_period == 0
        if kind == "periodic":
            g = R.period_disturb(draws[k], p, sk)
            f = g
            c = 0.0
        elif kind == "sin":
            g, c = R.sin_disturb(p, sk), 0.0
        elif kind == "mixed":
            g = (R.sin_disturb(p, sk) + (draws[k] if hit else 0.0)) / 3.0
            c = 0.0 if hit else 1.0 / 3.0
        else:
            g, c = np.zeros(3), 0.0
        tab[k + 1, :3], tab[k + 1, 3] = g, c
    return tab
