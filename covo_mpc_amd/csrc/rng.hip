// rng.hip -- counter-based standard-normal fill (Philox4x32-10 + Box-Muller), gfx950.
//
// Stands in for jax.random.split + jax.random.normal inside jax.random.multivariate_normal
// (quadjax/controllers/covo.py:212-220, mppi.py:53-65).  jax's threefry bitstream is unpinned
// (setup.py:21) and unavailable here, so the stream is build-defined; what is kept is the
// PROPERTY the sample-sharded step needs: element (global sample id, column) depends only on
// (key0, key1, id, column) -- never on the shard geometry.
//   counter = (column/4, id_lo, id_hi, 0), key = (key0, key1)  -> 4 x u32
//   u = ((x >> 8) + 0.5) * 2^-24 in (0,1);  z = sqrt(-2 ln u1) * {cos, sin}(2 pi u2)
#include "covo_common.hpp"

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4])
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

__global__ __launch_bounds__(256) void randn_kernel(uint32_t k0, uint32_t k1, int64_t sample_offset, int n_samples,
                                                    int n_cols4, float4 *__restrict__ out)
{
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)n_samples * n_cols4;
    if (gid >= total) return;
    const int c4 = (int)(gid % n_cols4);
    const uint64_t id = (uint64_t)(sample_offset + (int64_t)(gid / n_cols4));
    uint32_t r[4];
    philox4x32_10((uint32_t)c4, (uint32_t)id, (uint32_t)(id >> 32), 0u, k0, k1, r);
    const float ra = sqrtf(-2.0f * logf(u01(r[0]))), rb = sqrtf(-2.0f * logf(u01(r[2])));
    float sa, ca, sb, cb;
    sincosf(6.283185307179586f * u01(r[1]), &sa, &ca);
    sincosf(6.283185307179586f * u01(r[3]), &sb, &cb);
    out[gid] = make_float4(ra * ca, ra * sa, rb * cb, rb * sb);
}

int launch_randn(uint32_t k0, uint32_t k1, int64_t off, int n_samples, int n_cols, float *out, hipStream_t s)
{
    if (n_cols % 4 != 0) {
        covo_set_error("covo_randn: n_cols=%d must be a multiple of 4", n_cols);
        return COVO_E_BADARG;
    }
    const size_t total = (size_t)n_samples * (n_cols / 4);
    const int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(randn_kernel, dim3(grid), dim3(256), 0, s, k0, k1, off, n_samples, n_cols / 4,
                       reinterpret_cast<float4 *>(out));
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}
