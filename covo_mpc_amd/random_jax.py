"""jax.random's bitstream, restated: threefry2x32-20 keyed and laid out the way jax 0.4.x does it by default
(`jax_threefry_partitionable=False`), so that a JAX-equipped machine can replay the controller's noise without shipping
tensors: the epsilon this module (and `covo_randn_jax` on the device) produces for a controller key is what

    act_keys = jax.random.split(act_key, N)                               # controllers/covo.py:213
    eps_i    = jax.random.normal(act_keys[i], (H * du,))                  # inside multivariate_normal, covo.py:215-218

yields in quadjax (and, with keys split (N, H) and `normal(key, (du,))`, in mppi.py:53-65).  SURVEY.md 8f-4.

jax / jaxlib are not installed here and the reference leaves their version unpinned (setup.py:21), so nothing below can be
run against jax itself.  It is pinned by what IS published: the Random123 known-answer vectors of threefry2x32-20 (the same
three jax's own test-suite uses) and the values jax's documentation prints for `PRNGKey(0)`, `split(PRNGKey(0))` and
`normal(PRNGKey(0), (10,))` (tests/test_host.py).  Restated from the published algorithm, not copied:

  threefry2x32     Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11), 20 rounds, rotations
                   (13, 15, 26, 6) / (17, 29, 16, 24), key-schedule parity 0x1BD11BDA
  PRNGKey(seed)    key = (seed >> 32, seed & 0xffffffff)
  threefry_2x32(key, counts)   counts (even length, else one zero appended) is cut in two halves x0, x1; the block
                   function is applied to the pairs (x0[i], x1[i]); outputs are concatenated halves again
  split(key, n)    threefry_2x32(key, iota(2 n)).reshape(n, 2)
  random_bits(key, (m,))       threefry_2x32(key, iota(m))
  uniform(key, shape, minval, maxval)   f = bitcast((bits >> 9) | 0x3f800000) - 1;  max(minval, f (maxval - minval) + minval)
  normal(key, shape)           sqrt(2) erf_inv(uniform(key, shape, nextafter(-1, 0), 1)),  erf_inv = Giles' single-precision
                   polynomial (what XLA evaluates; its log1p may differ from numpy's in the last ulp)

The product's default stream stays Philox (random.py, csrc/rng_device.hpp): one block gives four 32-bit words for ~40 integer
operations, threefry two words for ~75, so drawing epsilon this way costs about as much as the whole rollout.
"""
from __future__ import annotations

import numpy as np

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_U32 = np.uint32


def _rotl(x, r):
    return (x << _U32(r)) | (x >> _U32(32 - r))


def threefry2x32(k0, k1, x0, x1):
    """Threefry-2x32, 20 rounds, on arrays of counter words.  -> (y0, y1) uint32 arrays."""
    with np.errstate(over="ignore"):
        x0 = np.array(x0, dtype=_U32, copy=True).reshape(-1)
        x1 = np.array(x1, dtype=_U32, copy=True).reshape(-1)
        ks = (_U32(k0), _U32(k1), _U32(k0) ^ _U32(k1) ^ _U32(0x1BD11BDA))
        x0 += ks[0]
        x1 += ks[1]
        for block in range(5):
            for r in _ROT[block & 1]:
                x0 += x1
                x1 = _rotl(x1, r) ^ x0
            x0 += ks[(block + 1) % 3]
            x1 += ks[(block + 2) % 3] + _U32(block + 1)
    return x0, x1


def _threefry_counts(key, counts):
    """jax's threefry_2x32(keypair, count): halves in, halves out."""
    counts = np.asarray(counts, dtype=_U32).reshape(-1)
    odd = counts.size % 2
    if odd:
        counts = np.concatenate([counts, np.zeros(1, dtype=_U32)])
    half = counts.size // 2
    y0, y1 = threefry2x32(key[0], key[1], counts[:half], counts[half:])
    out = np.concatenate([y0, y1])
    return out[:-1] if odd else out


def PRNGKey(seed: int) -> np.ndarray:
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return np.array([seed >> 32, seed & 0xFFFFFFFF], dtype=_U32)


def split(key, num: int = 2) -> np.ndarray:
    return _threefry_counts(key, np.arange(2 * num, dtype=_U32)).reshape(num, 2)


def random_bits(key, n: int) -> np.ndarray:
    return _threefry_counts(key, np.arange(n, dtype=_U32))


def uniform(key, shape=(), minval=0.0, maxval=1.0) -> np.ndarray:
    n = int(np.prod(shape)) if shape != () else 1
    f = ((random_bits(key, n) >> _U32(9)) | _U32(0x3F800000)).view(np.float32) - np.float32(1.0)
    lo, hi = np.float32(minval), np.float32(maxval)
    u = np.maximum(lo, f * (hi - lo) + lo)
    return u.reshape(shape) if shape != () else u[0]


def erf_inv_f32(x):
    """Giles, "Approximating the erfinv function" (GPU Computing Gems 2), single precision; evaluated in fp32."""
    x = np.asarray(x, dtype=np.float32)
    f = np.float32
    with np.errstate(divide="ignore", invalid="ignore"):
        w = -np.log1p(-x * x).astype(np.float32)
        small = w < f(5.0)
        ws = w - f(2.5)
        p = f(2.81022636e-08)
        for c in (3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087, -0.00125372503, -0.00417768164, 0.246640727,
                  1.50140941):
            p = f(c) + p * ws
        wl = np.sqrt(w) - f(3.0)
        q = f(-0.000200214257)
        for c in (0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773, -0.0076224613, 0.00943887047, 1.00167406,
                  2.83297682):
            q = f(c) + q * wl
        out = np.where(small, p, q) * x
    return np.where(np.abs(x) == f(1.0), np.copysign(f(np.inf), x), out).astype(np.float32)


def normal(key, shape=()) -> np.ndarray:
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    u = uniform(key, shape if shape != () else (1,), lo, 1.0)
    z = (np.float32(np.sqrt(2.0)) * erf_inv_f32(u)).astype(np.float32)
    return z if shape != () else z[0]


def controller_epsilon(act_key, N: int, n: int = 128, sample_offset: int = 0, n_samples: int | None = None) -> np.ndarray:
    """(n_samples, n) fp32: row i = jax.random.normal(jax.random.split(act_key, N)[sample_offset + i], (n,)) -- the
    standard-normal draws inside quadjax's vmapped multivariate_normal (covo.py:213-220)."""
    n_samples = N - sample_offset if n_samples is None else n_samples
    keys = split(act_key, N)[sample_offset:sample_offset + n_samples]
    out = np.empty((n_samples, n), dtype=np.float32)
    for i, k in enumerate(keys):
        out[i] = normal(k, (n,))
    return out


def controller_epsilon_mppi(act_key, N: int, H: int = 32, du: int = 4, sample_offset: int = 0, n_samples: int | None = None):
    """(n_samples, H * du) fp32: row i = concat_t normal(split(split(act_key, N)[i], H)[t], (du,)) -- mppi.py:53-60."""
    n_samples = N - sample_offset if n_samples is None else n_samples
    keys = split(act_key, N)[sample_offset:sample_offset + n_samples]
    out = np.empty((n_samples, H * du), dtype=np.float32)
    for i, k in enumerate(keys):
        for t, kt in enumerate(split(k, H)):
            out[i, du * t:du * t + du] = normal(kt, (du,))
    return out
