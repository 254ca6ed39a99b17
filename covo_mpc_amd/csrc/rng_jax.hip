// rng_jax.hip -- the controller's epsilon drawn from jax.random's bitstream (threefry2x32-20, jax 0.4.x default layout):
//   CoVO  (controllers/covo.py:213-220): key_i = split(act_key, N)[i];  eps[i][:] = normal(key_i, (128,))
//   MPPI  (controllers/mppi.py:53-60):   key_i = split(act_key, N)[i];  key_it = split(key_i, H)[t];  eps[i][4t:4t+4] = normal(key_it, (4,))
// Twin of covo_mpc_amd/random_jax.py (which is pinned to the Random123 vectors and to the values jax's documentation
// prints); SURVEY.md 8f-4.  jax itself is not available here, so this is validated against that twin only.
// One wave per sample: its 64 lanes hold the 64 threefry blocks of normal(key, (128,)) (block j -> columns j and 64 + j),
// or the 32 x 2 blocks of the per-step draws.  The integer stream is exact; the normals differ from a CPU jax by the last
// ulp of log1p / sqrt (XLA's and ocml's differ too).
#include "covo_common.hpp"

namespace {

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// Threefry-2x32, 20 rounds (Salmon et al., SC'11): key (k0, k1), counter (x0, x1) -> (x0, x1)
__device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t &x0, uint32_t &x1)
{
    const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
    x0 += ks[0];
    x1 += ks[1];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        if ((b & 1) == 0) {
            x0 += x1; x1 = rotl32(x1, 13) ^ x0;
            x0 += x1; x1 = rotl32(x1, 15) ^ x0;
            x0 += x1; x1 = rotl32(x1, 26) ^ x0;
            x0 += x1; x1 = rotl32(x1, 6) ^ x0;
        } else {
            x0 += x1; x1 = rotl32(x1, 17) ^ x0;
            x0 += x1; x1 = rotl32(x1, 29) ^ x0;
            x0 += x1; x1 = rotl32(x1, 16) ^ x0;
            x0 += x1; x1 = rotl32(x1, 24) ^ x0;
        }
        x0 += ks[(b + 1) % 3];
        x1 += ks[(b + 2) % 3] + (uint32_t)(b + 1);
    }
}

// word `w` (0 or 1) of child `i` of split(key, num): threefry_2x32(key, iota(2 num)).reshape(num, 2)[i][w].
// flat index f = 2 i + w; flat[f] = y0 of block f for f < num, y1 of block f - num otherwise; block m has counter (m, num + m).
__device__ __forceinline__ uint32_t split_word(uint32_t k0, uint32_t k1, uint64_t num, uint64_t i, int w)
{
    const uint64_t f = 2 * i + (uint64_t)w;
    const bool second = f >= num;
    const uint64_t m = second ? f - num : f;
    uint32_t x0 = (uint32_t)m, x1 = (uint32_t)(num + m);
    threefry2x32(k0, k1, x0, x1);
    return second ? x1 : x0;
}

// jax.random.normal's map of 32 random bits: uniform on [nextafter(-1, 0), 1) -> sqrt(2) erf_inv (Giles' polynomial)
__device__ __forceinline__ float normal_from_bits(uint32_t bits)
{
    const float f = __uint_as_float((bits >> 9) | 0x3F800000u) - 1.0f;
    const float lo = -0.99999994f;                         // nextafter(-1, 0)
    float u = __fadd_rn(__fmul_rn(f, __fsub_rn(1.0f, lo)), lo);  // no contraction: same roundings as the numpy twin
    u = fmaxf(lo, u);
    const float w = -log1pf(-__fmul_rn(u, u));
    float p;
    if (w < 5.0f) {
        const float t = __fsub_rn(w, 2.5f);
        p = 2.81022636e-08f;
        p = __fadd_rn(3.43273939e-07f, __fmul_rn(p, t));
        p = __fadd_rn(-3.5233877e-06f, __fmul_rn(p, t));
        p = __fadd_rn(-4.39150654e-06f, __fmul_rn(p, t));
        p = __fadd_rn(0.00021858087f, __fmul_rn(p, t));
        p = __fadd_rn(-0.00125372503f, __fmul_rn(p, t));
        p = __fadd_rn(-0.00417768164f, __fmul_rn(p, t));
        p = __fadd_rn(0.246640727f, __fmul_rn(p, t));
        p = __fadd_rn(1.50140941f, __fmul_rn(p, t));
    } else {
        const float t = __fsub_rn(sqrtf(w), 3.0f);
        p = -0.000200214257f;
        p = __fadd_rn(0.000100950558f, __fmul_rn(p, t));
        p = __fadd_rn(0.00134934322f, __fmul_rn(p, t));
        p = __fadd_rn(-0.00367342844f, __fmul_rn(p, t));
        p = __fadd_rn(0.00573950773f, __fmul_rn(p, t));
        p = __fadd_rn(-0.0076224613f, __fmul_rn(p, t));
        p = __fadd_rn(0.00943887047f, __fmul_rn(p, t));
        p = __fadd_rn(1.00167406f, __fmul_rn(p, t));
        p = __fadd_rn(2.83297682f, __fmul_rn(p, t));
    }
    return __fmul_rn(1.41421354f, __fmul_rn(p, u));  // sqrt(2) erf_inv(u)
}

// MPPI = false: eps[i][j], eps[i][64 + j] from block j of normal(key_i, (128,));  MPPI = true: lane = 2 t + b holds block b
// of normal(split(key_i, H)[t], (4,)) -> columns 4 t + b and 4 t + 2 + b
template <bool MPPI>
__global__ __launch_bounds__(256) void randn_jax_kernel(uint32_t k0, uint32_t k1, uint64_t n_total, uint64_t sample_offset,
                                                        int n_samples, float *__restrict__ eps)
{
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_samples) return;
    const uint64_t i = sample_offset + (uint64_t)s;
    // key_i = split(act_key, N)[i]: lanes compute word (lane & 1), lanes 0 and 1 hold the key
    const uint32_t kw = split_word(k0, k1, n_total, i, lane & 1);
    uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)kw, 0), c1 = (uint32_t)__builtin_amdgcn_readlane((int)kw, 1);
    float *row = eps + (size_t)s * COVO_NA;
    if (MPPI) {
        const int t = lane >> 1, b = lane & 1;
        const uint32_t s0 = split_word(c0, c1, COVO_H, (uint64_t)t, 0), s1 = split_word(c0, c1, COVO_H, (uint64_t)t, 1);
        uint32_t x0 = (uint32_t)b, x1 = (uint32_t)(2 + b);  // threefry_2x32(key, iota(4)): blocks (0, 2), (1, 3)
        threefry2x32(s0, s1, x0, x1);
        row[4 * t + b] = normal_from_bits(x0);
        row[4 * t + 2 + b] = normal_from_bits(x1);
    } else {
        uint32_t x0 = (uint32_t)lane, x1 = (uint32_t)(64 + lane);  // threefry_2x32(key, iota(128)): block j = (j, 64 + j)
        threefry2x32(c0, c1, x0, x1);
        row[lane] = normal_from_bits(x0);
        row[64 + lane] = normal_from_bits(x1);
    }
}

}  // namespace

int launch_randn_jax(uint32_t k0, uint32_t k1, int64_t n_total, int64_t off, int n_samples, int mppi, float *out, hipStream_t s)
{
    const int grid = (n_samples + 3) / 4;
    if (mppi)
        hipLaunchKernelGGL(randn_jax_kernel<true>, dim3(grid), dim3(256), 0, s, k0, k1, (uint64_t)n_total, (uint64_t)off, n_samples, out);
    else
        hipLaunchKernelGGL(randn_jax_kernel<false>, dim3(grid), dim3(256), 0, s, k0, k1, (uint64_t)n_total, (uint64_t)off, n_samples, out);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}
