"""TEST INFRASTRUCTURE (oracle): an independent restatement of jax.random's default bitstream, the checker of
`covo_randn_jax` (csrc/rng_jax.hip) and of the controllers' `noise_stream = "jax"` path.  Only tests/ may import it.

It restates the PUBLISHED algorithms, not the product's twin (covo_mpc_amd/random_jax.py): integers are carried in uint64 and
masked to 32 bits (the product wraps uint32); the inverse error function is Giles' single-precision polynomial pair (what XLA
evaluates: "Approximating the erfinv function", GPU Computing Gems Jade ed., ch. 10 -- coefficients as published) written as
two `np.polyval` calls in fp32, and bounded against scipy's double-precision `erfinv` (`normal_exact`: the fp32 formula loses
digits in the tails, where 1 - u^2 cancels -- up to 3e-5 at |z| > 4 --, which is what jax returns there too).  Pinned to what
jax publishes (tests/test_oracle.py: the Random123 threefry2x32-20 vectors jax's own tests use, and the values jax's
documentation prints for PRNGKey(0)); jax itself is not installed here (parity with it stays unpinned).

  threefry2x32-20   Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11 -- 5 groups of 4 rounds,
                    rotation constants 13 15 26 6 | 17 29 16 24, key word k2 = k0 ^ k1 ^ 0x1BD11BDA, subkey injection after
                    every group with the group counter added to the second word
  jax (0.4, `jax_threefry_partitionable` off):  threefry_2x32(key, counts) encrypts (counts[:h], counts[h:]) pairwise, h = half
                    the (zero-padded to even) length, and concatenates the two output halves;  split(key, n) = that on iota(2n),
                    viewed (n, 2);  bits(key, m) = that on iota(m);  uniform = bitcast(bits >> 9 | 0x3f800000) - 1 scaled;
                    normal = sqrt(2) erfinv(uniform on (nextafter(-1, 0), 1))
  quadjax           covo.py:213-220: eps_i = normal(split(act_key, N)[i], (H du,));  mppi.py:53-60: per step t
                    normal(split(split(act_key, N)[i], H)[t], (du,))
"""
import numpy as np
from scipy.special import erfinv

M32 = np.uint64(0xFFFFFFFF)
R_EVEN = (13, 15, 26, 6)
R_ODD = (17, 29, 16, 24)


def _rot(v, r):
    return ((v << np.uint64(r)) | (v >> np.uint64(32 - r))) & M32


def threefry_block(k0, k1, c0, c1):
    """One threefry2x32-20 block per element of (c0, c1) under the key (k0, k1).  -> two uint32 arrays."""
    k = [np.uint64(int(k0) & 0xFFFFFFFF), np.uint64(int(k1) & 0xFFFFFFFF)]
    k.append(k[0] ^ k[1] ^ np.uint64(0x1BD11BDA))
    a = (np.asarray(c0, dtype=np.uint64).reshape(-1) + k[0]) & M32
    b = (np.asarray(c1, dtype=np.uint64).reshape(-1) + k[1]) & M32
    for grp in range(5):
        for r in (R_EVEN if grp % 2 == 0 else R_ODD):
            a = (a + b) & M32
            b = _rot(b, r) ^ a
        a = (a + k[(grp + 1) % 3]) & M32
        b = (b + k[(grp + 2) % 3] + np.uint64(grp + 1)) & M32
    return a.astype(np.uint32), b.astype(np.uint32)


def _stream(key, counts):
    counts = np.asarray(counts, dtype=np.uint64).reshape(-1)
    n = counts.size
    if n % 2:
        counts = np.append(counts, np.uint64(0))
    h = counts.size // 2
    lo, hi = threefry_block(key[0], key[1], counts[:h], counts[h:])
    return np.concatenate([lo, hi])[:n]


def prng_key(seed):
    seed = int(seed) % (1 << 64)
    return np.array([seed >> 32, seed & 0xFFFFFFFF], dtype=np.uint32)


def split(key, n=2):
    return _stream(key, np.arange(2 * n)).reshape(n, 2)


def bits(key, m):
    return _stream(key, np.arange(m))


def uniform(key, m, lo=0.0, hi=1.0):
    f = ((bits(key, m) >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1)
    lo, hi = np.float32(lo), np.float32(hi)
    return np.maximum(lo, f * (hi - lo) + lo)


# Giles' coefficients, highest power first: the central branch in (w - 2.5), w = -log(1 - x^2) < 5, the tail branch in (sqrt(w) - 3)
GILES_CENTRAL = np.array([2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087, -0.00125372503,
                          -0.00417768164, 0.246640727, 1.50140941], dtype=np.float32)
GILES_TAIL = np.array([-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773, -0.0076224613,
                       0.00943887047, 1.00167406, 2.83297682], dtype=np.float32)


def erfinv_f32(x):
    x = np.asarray(x, dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = (-np.log1p(-(x * x))).astype(np.float32)
        central = np.polyval(GILES_CENTRAL, (w - np.float32(2.5)).astype(np.float32)).astype(np.float32)
        tail = np.polyval(GILES_TAIL, (np.sqrt(w) - np.float32(3)).astype(np.float32)).astype(np.float32)
        y = np.where(w < np.float32(5), central, tail) * x
    return np.where(np.abs(x) == 1, np.copysign(np.float32(np.inf), x), y).astype(np.float32)


def _normal_uniform(key, m):
    return uniform(key, m, np.nextafter(np.float32(-1), np.float32(0)), 1.0)


def normal(key, m):
    return (np.float32(np.sqrt(2.0)) * erfinv_f32(_normal_uniform(key, m))).astype(np.float32)


def normal_exact(key, m):
    """The same draw through double-precision erfinv: the yardstick of erfinv_f32 (agrees to fp32 rounding away from the tails)."""
    return np.sqrt(2.0) * erfinv(_normal_uniform(key, m).astype(np.float64))


def controller_epsilon(act_key, N, n=128, offset=0, count=None):
    """(count, n): row i = normal(split(act_key, N)[offset + i], (n,))  (covo.py:213-220)."""
    count = N - offset if count is None else count
    keys = split(act_key, N)[offset:offset + count]
    return np.stack([normal(k, n) for k in keys])


def controller_epsilon_mppi(act_key, N, H=32, du=4, offset=0, count=None):
    """(count, H du): row i = concat_t normal(split(split(act_key, N)[offset + i], H)[t], (du,))  (mppi.py:53-60)."""
    count = N - offset if count is None else count
    keys = split(act_key, N)[offset:offset + count]
    return np.stack([np.concatenate([normal(kt, du) for kt in split(k, H)]) for k in keys])
