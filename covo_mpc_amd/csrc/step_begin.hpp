// step_begin.hpp -- what a control step does before its first sample is drawn, as device functions shared by the begin launch
// of the staged step (step.hip: step_begin_kernel) and the fused small step (step_small.hip), which does the same per workgroup.
#pragma once
#include "covo_common.hpp"
#include "rng_device.hpp"

// MPPI's three tiny launches in one (mppi.py:43-49,59-61): shift the H covariance blocks in place (drop the first, repeat
// the last) and factor each 4x4 block -- thread t owns block t; same arithmetic as covo_cholesky (sigma.hip:
// symmetrise, fp64 right-looking Cholesky with sqrt and one division per column, fp32 out)
// (called by every thread of the launch: it contains a barrier)
// mppi_factor_block: the factor of ONE (already selected) 4x4 block, the arithmetic both callers share.  The contraction
// mode is pinned: translation units built with -ffp-contract=off (the rollout's) must form the same fused multiply-adds as step.hip.
__device__ __forceinline__ void mppi_factor_block(const float (&blk)[16], float *__restrict__ Lt)
{
#pragma clang fp contract(fast)
    double A[4][4];  // lower triangle, A[c][r] for r >= c (column-major like the LDS version)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = c; r < 4; ++r) A[c][r] = 0.5 * ((double)blk[4 * r + c] + (double)blk[4 * c + r]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double djj = sqrt(A[j][j]);
        const double inv = 1.0 / djj;
        A[j][j] = djj;
#pragma unroll
        for (int i = j + 1; i < 4; ++i) A[j][i] = A[j][i] * inv;
#pragma unroll
        for (int c = j + 1; c < 4; ++c)
#pragma unroll
            for (int i = c; i < 4; ++i) A[c][i] -= A[j][i] * A[j][c];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) Lt[4 * r + c] = (c <= r) ? (float)A[c][r] : 0.0f;
}

__device__ __forceinline__ void mppi_prep(float *__restrict__ a_cov, float *__restrict__ Ls)
{
    const int t = threadIdx.x;  // >= 32 threads, H = 32 active
    float blk[16];
    if (t < COVO_H) {
        const float *src = a_cov + 16 * ((t < COVO_H - 1) ? t + 1 : t);
#pragma unroll
        for (int i = 0; i < 16; ++i) blk[i] = src[i];
    }
    __syncthreads();  // every block is read before any is overwritten
    if (t >= COVO_H) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) a_cov[16 * t + i] = blk[i];
    mppi_factor_block(blk, Ls + 16 * t);
}

// child i of split(key, 2) / element i of normal(key, (3,)) exactly as covo_mpc_amd/random.py forms them
__device__ __forceinline__ void host_split(const uint32_t (&key)[2], uint32_t i, uint32_t (&child)[2])
{
    uint32_t r[4];
    rngd::philox4x32_10(i, 0u, 0u, 0x5EEDu, key[0], key[1], r);
    child[0] = r[0];
    child[1] = r[1];
}
__device__ __forceinline__ float host_normal3(const uint32_t (&key)[2], int i)
{
    uint32_t b1[4], b2[4];
    rngd::philox4x32_10(0u, 0u, 0u, 0xB175u, key[0], key[1], b1);
    rngd::philox4x32_10((uint32_t)((3 + i) >> 2), 0u, 0u, 0xB175u, key[0], key[1], b2);
    const double u1 = ((double)(b1[i] >> 8) + 0.5) / 16777216.0;
    const double u2 = ((double)(b2[(3 + i) & 3] >> 8) + 0.5) / 16777216.0;
    return (float)(sqrt(-2.0 * log(u1)) * cos(2.0 * 3.141592653589793 * u2));
}


// What changes per step travels in the begin launch's kernel arguments (48 bytes): the controller's raw rng_act, the caller's
// shared disturbance, the address of the state.
struct DynBlock {
    uint32_t w[12];  // {key0, key1, f_shared[3] as float bits, pad[3], state pointer (8 bytes), raw rng_act (device block only)}
};

// Virtual thread q of the per-step scalars (0: the sampling key, 1..3: the components of the shared disturbance) -> dyn[0..4]
// (+ dyn[10..11] = the raw controller key) exactly as step_begin_kernel has always formed them:
//   rng, act_key = split(rng_act); rng, step_key = split(rng)                (covo.py:212,225 / mppi.py:53,69)
//   MPPI: f_shared = scale * normal(split(split(split(step_key)[1])[0])[0], (3,))   (quadrotor.py:262, free.py:136,144)
__device__ __forceinline__ void step_begin_derive(const int q, const DynBlock &blk, const int derive_keys, const float shared_noise_scale,
                                                  uint32_t *__restrict__ dyn)
{
    const uint32_t raw[2] = {blk.w[0], blk.w[1]};
    {
        if (q == 0) {  // the raw controller key, for what else is derived from it in the graph (disturb.hip: the step's tables)
            dyn[10] = raw[0];
            dyn[11] = raw[1];
        }
        if (!derive_keys) {
            if (q == 0) {
                dyn[0] = raw[0];
                dyn[1] = raw[1];
            } else {
                dyn[2 + (q - 1)] = blk.w[2 + (q - 1)];
            }
            return;
        }
        uint32_t rng1[2], k[2], t[2];
        host_split(raw, 0u, rng1);
        if (q == 0) {
            host_split(raw, 1u, k);
            dyn[0] = k[0];
            dyn[1] = k[1];
        } else {
            float f = 0.0f;
            if (shared_noise_scale != 0.0f) {
                host_split(rng1, 1u, k);  // step_key
                host_split(k, 1u, t);     // raw_step: key, step_key = split(key)
                host_split(t, 0u, k);     // step_fn:  key, key_dyn = split(key)
                host_split(k, 0u, t);     // disturb_key, key = split(key)
                f = shared_noise_scale * host_normal3(t, q - 1);
            }
            dyn[2 + (q - 1)] = __float_as_uint(f);
        }
    }
}

