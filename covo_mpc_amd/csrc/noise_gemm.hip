// noise_gemm.hip -- correlated-noise draw  a = clip(mu + L eps, -1, 1)  as an fp32 MFMA GEMM (gfx950).
//
// Replaces jax.random.multivariate_normal's `mean + factor @ eps` and the following jnp.clip:
//   quadjax/controllers/covo.py:215-224 (full 128x128 factor)
//   quadjax/controllers/mppi.py:53-66   (32 independent 4x4 factors -> noise_blockdiag_kernel)
//
// out[128 x N] = L[128 x 128] . eps^T[128 x N] on v_mfma_f32_32x32x2_f32 (exact fp32: every dot
// product is an ascending-k fmaf chain, bit-identical to oracle_noise_gemm_f32).
//
// Structure (one wave = one 32-sample column tile x all 128 rows; waves are independent):
//   * L is lower-triangular: row tile rt (32 rows) only needs k < 32(rt+1) -> 160 of the 256
//     MFMAs per tile are issued (0.625 of dense; skipped terms are exact zeros).
//   * the 160 A-fragments of L live in VGPRs for the whole kernel (staged once per workgroup
//     through a padded LDS image, conflict-free ds_read_b32) -- the main loop touches no LDS.
//   * B fragments come straight from global memory: lanes (j, kh) of a wave load float4 chunk
//     2c+kh of sample row j (the lane pair consumes the whole 512-B row, full 128-B lines), and
//     two v_permlane32_swap per chunk pair put elements {k, k+1} on lanes {j, j+32} as the
//     32x32x2 B operand wants them.
//   * the D layout (col = lane&31 -> sample, rows 8g+4h+i -> (t, d)) makes every lane own whole
//     float4 actions, so the epilogue (+mu, clip) stores 512-B contiguous runs of the stripe
//     layout a[H][N][4] the rollout kernel reads.
// fp32 MFMA roofline: 2*128*128 flop per sample (dense-equivalent), 157.3 TFLOP/s peak.
#include "covo_common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NG_BLOCK = 256;
constexpr int NG_LDA = COVO_NA + 1;  // padded leading dimension of the LDS image of L

__device__ __forceinline__ constexpr int lf_off(int rt) { return 8 * rt * (rt + 1); }  // sum_{r<rt} 16(r+1)

struct BGroup {
    float4 c[4];  // this lane's chunks (2*(4g+i) + kh) of its sample row, i = 0..3
};

__device__ __forceinline__ BGroup load_group(const float4 *__restrict__ row, int g, int kh)
{
    BGroup b;
#pragma unroll
    for (int i = 0; i < 4; ++i) b.c[i] = row[2 * (4 * g + i) + kh];
    return b;
}

template <int G>
__device__ __forceinline__ void mfma_group(const float (&Lf)[160], BGroup b, f32x16 (&acc)[4])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float x = b.c[i].x, y = b.c[i].y, z = b.c[i].z, w = b.c[i].w;
        // lanes 32-63 of vdst <-> lanes 0-31 of src
        auto r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(z), __float_as_uint(w), false, false);
        const float bk0 = __uint_as_float(r0[0]);  // k = 8c+0 | 8c+1
        const float bk4 = __uint_as_float(r0[1]);  // k = 8c+4 | 8c+5
        const float bk2 = __uint_as_float(r1[0]);  // k = 8c+2 | 8c+3
        const float bk6 = __uint_as_float(r1[1]);  // k = 8c+6 | 8c+7
        const float bb[4] = {bk0, bk2, bk4, bk6};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            constexpr int dummy = 0;
            (void)dummy;
            const int ks = 16 * G + 4 * i + q;  // k-step (k0 = 2 ks)
#pragma unroll
            for (int rt = G; rt < 4; ++rt)
                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Lf[lf_off(rt) + ks], bb[q], acc[rt], 0, 0, 0);
        }
    }
}

__global__ __launch_bounds__(NG_BLOCK) void noise_gemm_kernel(const float *__restrict__ L, const float *__restrict__ mu,
                                                              const float *__restrict__ eps, int N, int ntiles,
                                                              float4 *__restrict__ a_out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ls = smem;                       // [128][NG_LDA]
    float *mus = smem + COVO_NA * NG_LDA;   // [128]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kh = lane >> 5;

    // ---- stage L (masked to its lower triangle) and mu
    for (int idx = tid; idx < COVO_NA * COVO_NA / 4; idx += NG_BLOCK) {
        const int i = idx >> 5, k4 = (idx & 31) * 4;
        const float4 v = reinterpret_cast<const float4 *>(L)[idx];
        float *d = Ls + i * NG_LDA + k4;
        d[0] = (k4 + 0 <= i) ? v.x : 0.0f;
        d[1] = (k4 + 1 <= i) ? v.y : 0.0f;
        d[2] = (k4 + 2 <= i) ? v.z : 0.0f;
        d[3] = (k4 + 3 <= i) ? v.w : 0.0f;
    }
    if (tid < COVO_NA) mus[tid] = mu[tid];
    __syncthreads();

    // ---- A fragments: lane supplies A[i = lane&31][k = 2 ks + (lane>>5)] for each 32-row tile
    float Lf[160];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ks = 0; ks < 16 * (rt + 1); ++ks)
            Lf[lf_off(rt) + ks] = Ls[(32 * rt + j) * NG_LDA + 2 * ks + kh];

    // this lane's mean entries: t = 8 rt + 2 g + kh
    // (read from LDS in the epilogue; 2 distinct addresses per wave -> broadcast)

    const int wave_global = blockIdx.x * (NG_BLOCK / 64) + wave;
    const int wave_stride = gridDim.x * (NG_BLOCK / 64);

    int tile = wave_global;
    if (tile >= ntiles) return;
    auto rowptr = [&](int t) {
        int row = t * 32 + j;
        row = row < N ? row : N - 1;
        return reinterpret_cast<const float4 *>(eps + (size_t)row * COVO_NA);
    };
    const float4 *row = rowptr(tile);
    BGroup cur = load_group(row, 0, kh);

    for (; tile < ntiles; tile += wave_stride) {
        f32x16 acc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rt][e] = 0.0f;

        const int next_tile = tile + wave_stride;
        const float4 *nrow = rowptr(next_tile < ntiles ? next_tile : tile);

        BGroup nxt = load_group(row, 1, kh);
        mfma_group<0>(Lf, cur, acc);
        cur = nxt;
        nxt = load_group(row, 2, kh);
        mfma_group<1>(Lf, cur, acc);
        cur = nxt;
        nxt = load_group(row, 3, kh);
        mfma_group<2>(Lf, cur, acc);
        cur = nxt;
        nxt = load_group(nrow, 0, kh);
        mfma_group<3>(Lf, cur, acc);
        cur = nxt;

        // ---- epilogue: + mu, clip, stripe-ordered float4 stores
        const int n = tile * 32 + j;
        if (n < N) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int t = 8 * rt + 2 * g + kh;
                    const float4 m4 = *reinterpret_cast<const float4 *>(mus + 4 * t);
                    float4 v;
                    v.x = qm::clip11_(m4.x + acc[rt][4 * g + 0]);
                    v.y = qm::clip11_(m4.y + acc[rt][4 * g + 1]);
                    v.z = qm::clip11_(m4.z + acc[rt][4 * g + 2]);
                    v.w = qm::clip11_(m4.w + acc[rt][4 * g + 3]);
                    a_out[(size_t)t * N + n] = v;
                }
        }
        row = nrow;
    }
}

// MPPI: a[t] = clip(mu[t] + Ls[t] eps[t]) with 4x4 lower factors; eps [N][H][4] -> a [H][N][4].
// HBM-bound streaming kernel (1 KiB per sample); fmaf chains in ascending k like the oracle.
__global__ __launch_bounds__(256) void noise_blockdiag_kernel(const float *__restrict__ Ls, const float *__restrict__ mu,
                                                              const float4 *__restrict__ eps, int N,
                                                              float4 *__restrict__ a_out)
{
    __shared__ float sL[COVO_H * 16];
    __shared__ float sm[COVO_NA];
    for (int i = threadIdx.x; i < COVO_H * 16; i += 256) sL[i] = Ls[i];
    if (threadIdx.x < COVO_NA) sm[threadIdx.x] = mu[threadIdx.x];
    __syncthreads();
    // thread -> (sample, t): consecutive threads take consecutive t of one sample (coalesced reads);
    // writes are 16-B scattered by t but each (t) stripe is filled by neighbouring blocks.
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)N * COVO_H;
    if (gid >= total) return;
    const int t = (int)(gid % COVO_H);
    const size_t n = gid / COVO_H;
    const float4 e = eps[gid];
    const float ev[4] = {e.x, e.y, e.z, e.w};
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = fmaf((k <= i) ? sL[t * 16 + i * 4 + k] : 0.0f, ev[k], acc);
        o[i] = qm::clip11_(sm[t * 4 + i] + acc);
    }
    a_out[(size_t)t * N + n] = make_float4(o[0], o[1], o[2], o[3]);
}

int launch_noise_gemm(const float *L, const float *mu, const float *eps, int N, float *a, hipStream_t s)
{
    const int ntiles = (N + 31) / 32;
    const int waves_per_block = NG_BLOCK / 64;
    int grid = (ntiles + waves_per_block - 1) / waves_per_block;
    if (grid > 256) grid = 256;  // persistent: one workgroup per CU, waves stride over tiles
    const size_t lds = (size_t)(COVO_NA * NG_LDA + COVO_NA) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        COVO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(noise_gemm_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(noise_gemm_kernel, dim3(grid), dim3(NG_BLOCK), lds, s, L, mu, eps, N, ntiles,
                       reinterpret_cast<float4 *>(a));
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_noise_blockdiag(const float *Ls, const float *mu, const float *eps, int N, float *a, hipStream_t s)
{
    const size_t total = (size_t)N * COVO_H;
    const int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(noise_blockdiag_kernel, dim3(grid), dim3(256), 0, s, Ls, mu,
                       reinterpret_cast<const float4 *>(eps), N, reinterpret_cast<float4 *>(a));
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}
