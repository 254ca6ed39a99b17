"""Philox4x32-10 + Box-Muller in numpy.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Twin of covo_mpc_amd/csrc/rng.hip.  The integer stream is pinned by the Random123
known-answer vectors (tests/test_oracle.py); it is NOT jax's threefry stream (unpinned,
unavailable) -- epsilon is an explicit input of every parity interface instead.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _u01(x):
    return ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def randn(key0, key1, sample_offset, n_samples, n_cols):
    """float32 (n_samples, n_cols); element (id, col) depends only on (key, id, col)."""
    assert n_cols % 4 == 0
    ids = np.arange(n_samples, dtype=np.uint64) + np.uint64(sample_offset)
    c4 = np.arange(n_cols // 4, dtype=np.uint32)
    C0 = np.broadcast_to(c4[None, :], (n_samples, n_cols // 4))
    C1 = np.broadcast_to((ids & MASK).astype(np.uint32)[:, None], C0.shape)
    C2 = np.broadcast_to((ids >> np.uint64(32)).astype(np.uint32)[:, None], C0.shape)
    r0, r1, r2, r3 = philox4x32_10(C0, C1, C2, np.zeros_like(C0), key0, key1)
    two_pi = np.float32(6.283185307179586)
    ra = np.sqrt(np.float32(-2.0) * np.log(_u01(r0)))
    rb = np.sqrt(np.float32(-2.0) * np.log(_u01(r2)))
    ta = two_pi * _u01(r1)
    tb = two_pi * _u01(r3)
    out = np.stack([ra * np.cos(ta), ra * np.sin(ta), rb * np.cos(tb), rb * np.sin(tb)], axis=-1)
    return out.reshape(n_samples, n_cols).astype(np.float32)
