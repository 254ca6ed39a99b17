// reduce.hip -- exp(-cost/lambda)-weighted aggregation as a two-stage wavefront reduction (gfx950).
//
// Replaces quadjax/controllers/covo.py:266-278 (mppi.py:109-129):
//     w = exp(-(cost - min cost)/lam) / sum ;  a_mean' = gamma * sum_n w_n a_n + (1-gamma) a_mean
//
// Stage 1 (softmax_partial_kernel): every workgroup first reduces the per-wave (64-sample) cost minima left
//   by the rollout kernel (4 KiB at N = 65536, L2-resident) to the exact global minimum m, so weights are
//   formed exactly like the reference's `cost - jnp.min(cost)`.  A wave then walks 64-sample
//   groups: one coalesced cost load, w = exp((m-c)/lam); 8-sample sub-groups whose weights are
//   all exactly 0 (the overwhelming majority at lam = 0.01: exp underflows once c-m > 1.04) are
//   skipped -- their contribution is an exact zero.  For a live sub-group the wave reads the
//   stripes a[t][8 samples][4] as full 128-B lines (lane = (t mod 8, sample)), 4 loads cover all
//   32 steps, accumulating float4 partial sums; 3 xor-shuffle steps fold the 8 sample lanes.
//   Cross-wave fold through LDS; one {m, s, v[128]} record per workgroup (no atomics ->
//   bit-reproducible).
// Stage 2 (merge_kernel): one workgroup merges records with the online-softmax rule
//   (m = min m_g, scale_g = exp(-(m_g - m)/lam)); the same kernel merges the all-gathered
//   records of the G ranks of a sample-sharded step and applies the gamma blend.
// HBM roofline: 516 B/sample algorithmic (cost + stripes); far less is actually fetched when
// sub-groups are skipped.
#include "covo_common.hpp"

constexpr int RD_BLOCK = 256;
constexpr int RD_WAVES = RD_BLOCK / 64;

__global__ __launch_bounds__(256) void groupmin_kernel(const float *__restrict__ cost, int N, float *__restrict__ gm)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    const float wm = wave_min(n < N ? cost[n] : __builtin_inff());
    if ((threadIdx.x & 63) == 0 && (n >> 6) < (N + 63) / 64) gm[n >> 6] = wm;
}

__global__ __launch_bounds__(RD_BLOCK) void softmax_partial_kernel(const float *__restrict__ cost,
                                                                   const float4 *__restrict__ a, int N,
                                                                   const float *__restrict__ blockmin, int nbm,
                                                                   float inv_lam, float *__restrict__ partials)
{
    __shared__ float red[RD_WAVES];
    __shared__ float sv[RD_WAVES][COVO_NA];
    __shared__ float ss[RD_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // blockIdx.y (env-batched step): instance y's dense slices; its records follow those of instance y - 1
        const size_t y = blockIdx.y;
        cost += y * N;
        a += y * ((size_t)COVO_H * N);
        blockmin += y * nbm;
        partials += y * gridDim.x * COVO_PARTIAL_FLOATS;
    }

    // ---- exact global minimum of cost from the per-block minima
    float m = __builtin_inff();
    for (int i = tid; i < nbm; i += RD_BLOCK) m = fminf(m, blockmin[i]);
    m = wave_min(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));

    const int ngroups = (N + 63) / 64;
    const int sub = lane & 7, tq = lane >> 3;
    float4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    float s_lane = 0.0f;

    for (int g = blockIdx.x * RD_WAVES + wave; g < ngroups; g += gridDim.x * RD_WAVES) {
        const int n = g * 64 + lane;
        const float c = (n < N) ? cost[n] : __builtin_inff();
        const float w = expf((m - c) * inv_lam);  // covo.py:266
        s_lane += w;
        const unsigned long long live = __ballot(w > 0.0f);
        if (live == 0ull) continue;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (((live >> (8 * q)) & 0xffull) == 0ull) continue;  // wave-uniform
            const float wv = __shfl(w, 8 * q + sub, 64);
            int ns = g * 64 + 8 * q + sub;
            ns = ns < N ? ns : N - 1;  // wv == 0 there
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
                const float4 av = a[(size_t)(8 * tb + tq) * N + ns];
                acc[tb].x = fmaf(wv, av.x, acc[tb].x);
                acc[tb].y = fmaf(wv, av.y, acc[tb].y);
                acc[tb].z = fmaf(wv, av.z, acc[tb].z);
                acc[tb].w = fmaf(wv, av.w, acc[tb].w);
            }
        }
    }
    // fold the 8 sample lanes (lane bits 0..2)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            acc[tb].x += __shfl_xor(acc[tb].x, o, 64);
            acc[tb].y += __shfl_xor(acc[tb].y, o, 64);
            acc[tb].z += __shfl_xor(acc[tb].z, o, 64);
            acc[tb].w += __shfl_xor(acc[tb].w, o, 64);
        }
    }
    const float s_wave = wave_sum(s_lane);
    if (sub == 0) {
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) *reinterpret_cast<float4 *>(&sv[wave][4 * (8 * tb + tq)]) = acc[tb];
    }
    if (lane == 0) ss[wave] = s_wave;
    __syncthreads();
    float *rec = partials + (size_t)blockIdx.x * COVO_PARTIAL_FLOATS;
    if (tid < COVO_NA) rec[2 + tid] = (sv[0][tid] + sv[1][tid]) + (sv[2][tid] + sv[3][tid]);
    if (tid == 0) {
        rec[0] = m;
        rec[1] = (ss[0] + ss[1]) + (ss[2] + ss[3]);
    }
}

// Merges G records {m, s, v[128]} with 1024 threads = 8 record-slices x 128 columns.
// FINAL: a_mean_out = gamma * v/s + (1-gamma) * a_mean_old (covo.py:270-275); otherwise writes the
// merged record to out.  Fixed summation order -> bit-reproducible.
constexpr int MG_THREADS = 1024;
constexpr int MG_SLICES = MG_THREADS / COVO_NA;  // 8
constexpr int MG_MAXG = 1024;
// stride: floats between consecutive records (COVO_PARTIAL_FLOATS, or COVO_RANK_RECORD_FLOATS for the all-gathered rank records
// that also carry the position sums)
template <bool FINAL>
__global__ __launch_bounds__(MG_THREADS) void merge_kernel(const float *__restrict__ partials, int G, float inv_lam,
                                                           const float *__restrict__ a_mean_old, float gamma_mean,
                                                           float *__restrict__ out, int stride)
{
    __shared__ float scale[MG_MAXG];
    __shared__ float redm[MG_THREADS / 64];
    __shared__ float reds[MG_THREADS / 64];
    __shared__ float sv[MG_SLICES][COVO_NA];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // blockIdx.x (env-batched step): instance x merges its own G records into its own mean
        const size_t x = blockIdx.x;
        partials += x * G * stride;
        if (FINAL) a_mean_old += x * COVO_NA;
        out += x * (FINAL ? COVO_NA : COVO_PARTIAL_FLOATS);
    }
    // phase 1: m = min_g m_g
    float m = __builtin_inff();
    for (int g = tid; g < G; g += MG_THREADS) m = fminf(m, partials[(size_t)g * stride]);
    m = wave_min(m);
    if (lane == 0) redm[wave] = m;
    __syncthreads();
    m = redm[0];
#pragma unroll
    for (int i = 1; i < MG_THREADS / 64; ++i) m = fminf(m, redm[i]);
    // phase 2: per-record scale and s = sum_g s_g scale_g
    float s = 0.0f;
    for (int g = tid; g < G; g += MG_THREADS) {
        const float *rec = partials + (size_t)g * stride;
        const float sg = rec[1];
        const float sc = (sg > 0.0f) ? expf((m - rec[0]) * inv_lam) : 0.0f;  // empty shard -> 0
        scale[g] = sc;
        s = fmaf(sg, sc, s);
    }
    s = wave_sum(s);
    if (lane == 0) reds[wave] = s;
    __syncthreads();
    s = 0.0f;
#pragma unroll
    for (int i = 0; i < MG_THREADS / 64; ++i) s += reds[i];
    // phase 3: v[col] = sum_g v_g[col] scale_g, record slices in parallel
    const int col = tid & (COVO_NA - 1), slice = tid >> 7;
    float v = 0.0f;
#pragma unroll 4
    for (int g = slice; g < G; g += MG_SLICES) v = fmaf(partials[(size_t)g * stride + 2 + col], scale[g], v);
    sv[slice][col] = v;
    __syncthreads();
    if (tid < COVO_NA) {
        v = 0.0f;
#pragma unroll
        for (int i = 0; i < MG_SLICES; ++i) v += sv[i][tid];
        if (FINAL) {
            out[tid] = (v / s) * gamma_mean + a_mean_old[tid] * (1.0f - gamma_mean);
        } else {
            out[2 + tid] = v;
            if (tid == 0) {
                out[0] = m;
                out[1] = s;
            }
        }
    }
}

__global__ void shift_mean_kernel(const float *__restrict__ in, float *__restrict__ out)
{
    const int i = threadIdx.x;  // 128 threads; covo.py:201-203
    out[i] = (i < COVO_NA - COVO_DU) ? in[i + COVO_DU] : in[i];
}

int launch_softmax_reduce(covo_ctx *h, const float *cost, const float *a, int N, const float *blockmin, int n_blockmin,
                          float *partial_out, const float *a_mean_old, float gamma_mean, float *a_mean_out,
                          hipStream_t s, float *partials_ws, int batch)
{
    const float inv_lam = 1.0f / h->cfg.lam;
    if (partials_ws == nullptr) partials_ws = h->ws_partials;
    if (blockmin == nullptr) {
        n_blockmin = (N + 63) / 64;
        hipLaunchKernelGGL(groupmin_kernel, dim3((N + 255) / 256), dim3(256), 0, s, cost, N, h->ws_blockmin);
        blockmin = h->ws_blockmin;
    }
    const int ngroups = (N + 63) / 64;
    int grid = (ngroups + RD_WAVES - 1) / RD_WAVES;
    if (grid > h->max_red_blocks) grid = h->max_red_blocks;
    hipLaunchKernelGGL(softmax_partial_kernel, dim3(grid, batch), dim3(RD_BLOCK), 0, s, cost,
                       reinterpret_cast<const float4 *>(a), N, blockmin, n_blockmin, inv_lam, partials_ws);
    if (a_mean_out != nullptr)
        hipLaunchKernelGGL(merge_kernel<true>, dim3(batch), dim3(MG_THREADS), 0, s, partials_ws, grid, inv_lam, a_mean_old,
                           gamma_mean, a_mean_out, COVO_PARTIAL_FLOATS);
    else
        hipLaunchKernelGGL(merge_kernel<false>, dim3(batch), dim3(MG_THREADS), 0, s, partials_ws, grid, inv_lam,
                           (const float *)nullptr, 1.0f, partial_out, COVO_PARTIAL_FLOATS);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_merge(const float *partials, int G, float lam, const float *a_mean_old, float gamma_mean, float *a_mean_out,
                 hipStream_t s, float *partial_out, int batch, int stride)
{
    if (G > MG_MAXG) { covo_set_error("covo_merge: G=%d > %d", G, MG_MAXG); return COVO_E_BADARG; }
    if (a_mean_out != nullptr)
        hipLaunchKernelGGL(merge_kernel<true>, dim3(batch), dim3(MG_THREADS), 0, s, partials, G, 1.0f / lam, a_mean_old, gamma_mean,
                           a_mean_out, stride);
    else
        hipLaunchKernelGGL(merge_kernel<false>, dim3(batch), dim3(MG_THREADS), 0, s, partials, G, 1.0f / lam,
                           (const float *)nullptr, 1.0f, partial_out, stride);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_shift_mean(const float *in, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(shift_mean_kernel, dim3(1), dim3(COVO_NA), 0, s, in, out);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}
