"""Host-side PRNG keys mirroring the way quadjax threads `jax.random` keys through its call stack.

jax's threefry bitstream is unpinned (setup.py:21) and not reproducible here, so keys are
Philox4x32-10 counters: a key is two uint32 words; `split` derives children; draws are numpy
arrays (host plumbing: env reset, observation noise, trajectory generation).  The device-side
sampling noise uses the same Philox core (csrc/rng.hip) keyed by (key, global sample id, column).
"""
from __future__ import annotations

import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def _philox(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0, k1 = int(k0), int(k1)
    for _ in range(10):
        p0 = _M0 * c0.astype(np.uint64)
        p1 = _M1 * c2.astype(np.uint64)
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK).astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK).astype(np.uint32)
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint32(k0), lo1, hi0 ^ c3 ^ np.uint32(k1), lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def _philox_scalar(c0, c1, c2, c3, k0, k1):
    """Same function on Python ints (key splitting sits on the per-control-step host path)."""
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = (p1 >> 32) ^ c1 ^ k0, p1 & 0xFFFFFFFF, (p0 >> 32) ^ c3 ^ k1, p0 & 0xFFFFFFFF
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def PRNGKey(seed: int) -> np.ndarray:
    """jax.random.PRNGKey(seed) analogue: (hi, lo) words of the seed."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return np.array([seed >> 32, seed & 0xFFFFFFFF], dtype=np.uint32)


def split(key, num: int = 2) -> np.ndarray:
    """jax.random.split analogue: (num, 2) child keys."""
    k0, k1 = int(key[0]), int(key[1])
    if num <= 16:
        return np.array([_philox_scalar(i, 0, 0, 0x5EED, k0, k1)[:2] for i in range(num)], dtype=np.uint32)
    i = np.arange(num, dtype=np.uint32)
    z = np.zeros(num, dtype=np.uint32)
    r0, r1, _, _ = _philox(i, z, z, z + np.uint32(0x5EED), k0, k1)
    return np.stack([r0, r1], axis=-1)


def _bits(key, n: int) -> np.ndarray:
    m = (n + 3) // 4
    if m <= 8:  # scalar path: per-control-step draws (3-vectors) sit on the host critical path
        k0, k1 = int(key[0]), int(key[1])
        out = []
        for i in range(m):
            out.extend(_philox_scalar(i, 0, 0, 0xB175, k0, k1))
        return np.asarray(out[:n], dtype=np.uint32)
    i = np.arange(m, dtype=np.uint32)
    z = np.zeros(m, dtype=np.uint32)
    r = _philox(i, z, z, z + np.uint32(0xB175), key[0], key[1])
    return np.stack(r, axis=-1).reshape(-1)[:n]


def uniform(key, shape=(), minval=0.0, maxval=1.0, dtype=np.float32) -> np.ndarray:
    n = int(np.prod(shape)) if shape != () else 1
    u = ((_bits(key, n) >> np.uint32(8)).astype(np.float64) + 0.5) / 16777216.0
    out = (minval + (maxval - minval) * u).astype(dtype)
    return out.reshape(shape) if shape != () else out[0]


def normal(key, shape=(), dtype=np.float32) -> np.ndarray:
    n = int(np.prod(shape)) if shape != () else 1
    if n <= 4 and shape != ():  # scalar path (same formula in Python floats): the MPPI disturbance 3-vector
        import math
        k0, k1 = int(key[0]), int(key[1])
        b = []
        for i in range((2 * n + 3) // 4):
            b.extend(_philox_scalar(i, 0, 0, 0xB175, k0, k1))
        z = [math.sqrt(-2.0 * math.log(((b[i] >> 8) + 0.5) / 16777216.0)) *
             math.cos(2.0 * math.pi * (((b[n + i] >> 8) + 0.5) / 16777216.0)) for i in range(n)]
        return np.asarray(z, dtype=dtype).reshape(shape)
    b = _bits(key, 2 * n)
    u1 = ((b[:n] >> np.uint32(8)).astype(np.float64) + 0.5) / 16777216.0
    u2 = ((b[n:] >> np.uint32(8)).astype(np.float64) + 0.5) / 16777216.0
    z = (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(dtype)
    return z.reshape(shape) if shape != () else z[0]
