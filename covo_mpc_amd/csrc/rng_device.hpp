// rng_device.hpp -- Philox4x32-10 + Box-Muller device functions shared by randn_kernel (rng.hip) and
// the in-GEMM noise generation (noise_gemm.hip), so both produce bit-identical epsilon.
//   element (global sample id, column): counter = (column/4, id_lo, id_hi, 0), key = (key0, key1)
//   u = ((x >> 8) + 0.5) * 2^-24 in (0,1);  z = sqrt(-2 ln u1) * {cos, sin}(2 pi u2)
// Transcendentals are the gfx950 hardware ops: v_log_f32 (log2, 1 ulp), v_sqrt_f32 (1 ulp) and
// v_cos_f32 / v_sin_f32, which take their argument in REVOLUTIONS -- cos(2 pi u) is one instruction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rngd {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4])
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // the full 64-bit products: ONE v_mad_u64_u32 each instead of a v_mul_hi_u32 / v_mul_lo_u32 pair (all quarter rate)
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// the four normals of columns 4*c4 .. 4*c4+3 of global sample `id`
__device__ __forceinline__ float4 normal4(uint32_t c4, uint64_t id, uint32_t k0, uint32_t k1)
{
    uint32_t r[4];
    philox4x32_10(c4, (uint32_t)id, (uint32_t)(id >> 32), 0u, k0, k1, r);
    // -2 ln u = (-2 ln 2) log2 u
    const float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(r[0])));
    const float rb = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(r[2])));
    const float ua = u01(r[1]), ub = u01(r[3]);
    return make_float4(ra * __builtin_amdgcn_cosf(ua), ra * __builtin_amdgcn_sinf(ua), rb * __builtin_amdgcn_cosf(ub),
                       rb * __builtin_amdgcn_sinf(ub));
}

}  // namespace rngd
