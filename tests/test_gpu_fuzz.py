"""A fixed-seed slice of scripts/fuzz_parity.py in the suite: rollout (+ position statistics) against the fp64 oracle at random
(time, seed, N), ragged fused steps, the Sigma chain on random spectra (bottom multiplicity / gap / width / null blocks, batch 1 and
batched) against LAPACK, the adjoint Hessian against the C oracle's hyper-dual one."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomised_parity_sweep():
    env = dict(os.environ)
    for k in ("COVO_GRAPH", "COVO_NO_GRAPH"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "11", "10", "4", "12", "6"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-2000:]
    assert "FAIL" not in text, text[-2000:]
    assert "rollout: 10 passed" in text and "fused: 4 passed" in text and "sigma: 24 passed" in text and "hessian: 6 passed" in text, text[-1500:]
