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
#include <cstdlib>
#include <cstring>
#include "covo_common.hpp"
#include "softmax_merge.hpp"

constexpr int RD_BLOCK = 256;
constexpr int RD_WAVES = RD_BLOCK / 64;

__global__ __launch_bounds__(256) void groupmin_kernel(const float *__restrict__ cost, int N, float *__restrict__ gm)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    const float wm = wave_min(n < N ? cost[n] : __builtin_inff());
    if ((threadIdx.x & 63) == 0 && (n >> 6) < (N + 63) / 64) gm[n >> 6] = wm;
}

// COV (MPPI's covariance adaptation, mppi.py:119-125): the record also carries the weighted second moments of d = a - mu
// about the SHIFTED OLD mean mu (known before sampling; d is the clipped L eps, so no cancellation against mean^2):
// rec[COVO_PARTIAL_FLOATS + 10 t + j] = sum_n w_n d_i d_j for the 10 pairs i <= j of step t (cov_pair below).
constexpr int RD_COV_FLOATS = COVO_H * 10;                                   // 320
constexpr int RD_COV_RECORD_FLOATS = COVO_PARTIAL_FLOATS + RD_COV_FLOATS;    // 452
template <bool COV>
__global__ __launch_bounds__(RD_BLOCK) void softmax_partial_kernel(const float *__restrict__ cost,
                                                                   const float4 *__restrict__ a, int N,
                                                                   const float *__restrict__ blockmin, int nbm,
                                                                   float inv_lam, float *__restrict__ partials,
                                                                   const float4 *__restrict__ mu)
{
    constexpr int REC = COV ? RD_COV_RECORD_FLOATS : COVO_PARTIAL_FLOATS;
    __shared__ float red[RD_WAVES];
    __shared__ float sv[RD_WAVES][COVO_NA];
    __shared__ float sv2[COV ? RD_WAVES : 1][COV ? RD_COV_FLOATS : 1];
    __shared__ float ss[RD_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // blockIdx.y (env-batched step): instance y's dense slices; its records follow those of instance y - 1
        const size_t y = blockIdx.y;
        cost += y * N;
        a += y * ((size_t)COVO_H * N);
        blockmin += y * nbm;
        partials += y * gridDim.x * REC;
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
    float acc2[COV ? 4 : 1][10];  // COV: pairs (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2) (2,3) (3,3) of steps 8 tb + tq
    float4 mu4[COV ? 4 : 1];
    if (COV) {
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            mu4[tb] = mu[8 * tb + tq];
#pragma unroll
            for (int j = 0; j < 10; ++j) acc2[tb][j] = 0.0f;
        }
    }

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
                if (COV) {
                    const float d0 = av.x - mu4[tb].x, d1 = av.y - mu4[tb].y, d2 = av.z - mu4[tb].z, d3 = av.w - mu4[tb].w;
                    const float w0 = wv * d0, w1 = wv * d1, w2 = wv * d2, w3 = wv * d3;
                    acc2[tb][0] = fmaf(w0, d0, acc2[tb][0]);
                    acc2[tb][1] = fmaf(w0, d1, acc2[tb][1]);
                    acc2[tb][2] = fmaf(w0, d2, acc2[tb][2]);
                    acc2[tb][3] = fmaf(w0, d3, acc2[tb][3]);
                    acc2[tb][4] = fmaf(w1, d1, acc2[tb][4]);
                    acc2[tb][5] = fmaf(w1, d2, acc2[tb][5]);
                    acc2[tb][6] = fmaf(w1, d3, acc2[tb][6]);
                    acc2[tb][7] = fmaf(w2, d2, acc2[tb][7]);
                    acc2[tb][8] = fmaf(w2, d3, acc2[tb][8]);
                    acc2[tb][9] = fmaf(w3, d3, acc2[tb][9]);
                }
            }
        }
    }
    if (COV) {
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int j = 0; j < 10; ++j) {
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) acc2[tb][j] += __shfl_xor(acc2[tb][j], o, 64);
                if (sub == 0) sv2[wave][10 * (8 * tb + tq) + j] = acc2[tb][j];
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
    float *rec = partials + (size_t)blockIdx.x * REC;
    if (tid < COVO_NA) rec[2 + tid] = (sv[0][tid] + sv[1][tid]) + (sv[2][tid] + sv[3][tid]);
    if (COV) {
        for (int i = tid; i < RD_COV_FLOATS; i += RD_BLOCK)
            rec[COVO_PARTIAL_FLOATS + i] = (sv2[0][i] + sv2[1][i]) + (sv2[2][i] + sv2[3][i]);
    }
    if (tid == 0) {
        rec[0] = m;
        rec[1] = (ss[0] + ss[1]) + (ss[2] + ss[3]);
    }
}


// index of the pair (i, j), i <= j, in a record's 10 second moments per step
__device__ __forceinline__ int cov_pair(int i, int j)
{
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    return lo * 4 - lo * (lo - 1) / 2 + (hi - lo);
}

// MPPI with gamma_sigma != 0 (mppi.py:109-125): merge G stage-1 records that carry second moments, new mean as merge_kernel,
// then a_cov'[t] = gamma_sigma sum_n w_n (a_n - mean')(a_n - mean')^T + (1 - gamma_sigma) a_cov[t] with the NEW mean (sic):
// with d = a - mu, e = mean' - mu, m1 = sum w d:  sum w (d - e)(d - e)^T = S2 - m1 e^T - e m1^T + e e^T   (sum w = 1).
// FINAL = false (a sample-sharded rank, round 4): the merged, UNNORMALISED record {m, s, v[128], pad[2], S2[320]} goes to
// a_mean_out instead -- this rank's contribution to the exchange; the G rank records are then merged by the FINAL variant on
// every rank (the second moments are about mu, the shifted OLD mean, which all ranks share).
// stride: floats between consecutive records (RD_COV_RECORD_FLOATS, or COVO_RANK_RECORD_COV_FLOATS for all-gathered rank records).
template <bool FINAL>
__global__ __launch_bounds__(MG_THREADS) void merge_cov_kernel(const float *__restrict__ partials, int G, float inv_lam,
                                                               const float *__restrict__ a_mean_old, float gamma_mean,
                                                               const float *__restrict__ a_cov_old, float gamma_sigma,
                                                               float *__restrict__ a_mean_out, float *__restrict__ a_cov_out,
                                                               int stride)
{
    __shared__ float scale[MG_MAXG];
    __shared__ float redm[MG_THREADS / 64];
    __shared__ float reds[MG_THREADS / 64];
    __shared__ float sv[MG_SLICES][COVO_NA];
    __shared__ float sv2[MG_SLICES][RD_COV_FLOATS];
    __shared__ float smean[COVO_NA], sm1[COVO_NA], se[COVO_NA], s2[RD_COV_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int REC = stride;
    float m = __builtin_inff();
    for (int g = tid; g < G; g += MG_THREADS) m = fminf(m, partials[(size_t)g * REC]);
    m = wave_min(m);
    if (lane == 0) redm[wave] = m;
    __syncthreads();
    m = redm[0];
#pragma unroll
    for (int i = 1; i < MG_THREADS / 64; ++i) m = fminf(m, redm[i]);
    float s = 0.0f;
    for (int g = tid; g < G; g += MG_THREADS) {
        const float *rec = partials + (size_t)g * REC;
        const float sg = rec[1];
        const float sc = (sg > 0.0f) ? expf((m - rec[0]) * inv_lam) : 0.0f;
        scale[g] = sc;
        s = fmaf(sg, sc, s);
    }
    s = wave_sum(s);
    if (lane == 0) reds[wave] = s;
    __syncthreads();
    s = 0.0f;
#pragma unroll
    for (int i = 0; i < MG_THREADS / 64; ++i) s += reds[i];
    const int col = tid & (COVO_NA - 1), slice = tid >> 7;
    // one workgroup reads all G records (G x 1.8 KB): the loads of a thread's G / 8 records are independent, only the fmaf chain
    // is ordered -- unrolled so that eight are in flight (rolled, every trip paid its own memory round trip: 37 us at G = 256)
    float v = 0.0f;
#pragma unroll 8
    for (int g = slice; g < G; g += MG_SLICES) v = fmaf(partials[(size_t)g * REC + 2 + col], scale[g], v);
    sv[slice][col] = v;
    {
        // the 320 second-moment columns: columns col, col + 128 and (col < 64) col + 256 side by side
        const bool third = col + 2 * COVO_NA < RD_COV_FLOATS;
        float va = 0.0f, vb = 0.0f, vc = 0.0f;
#pragma unroll 4
        for (int g = slice; g < G; g += MG_SLICES) {
            const float *rec = partials + (size_t)g * REC + COVO_PARTIAL_FLOATS + col;
            const float sc = scale[g];
            va = fmaf(rec[0], sc, va);
            vb = fmaf(rec[COVO_NA], sc, vb);
            if (third) vc = fmaf(rec[2 * COVO_NA], sc, vc);
        }
        sv2[slice][col] = va;
        sv2[slice][col + COVO_NA] = vb;
        if (third) sv2[slice][col + 2 * COVO_NA] = vc;
    }
    __syncthreads();
    if (!FINAL) {  // the merged record, unnormalised (a_mean_out = record [RD_COV_RECORD_FLOATS])
        float *__restrict__ rec = a_mean_out;
        if (tid < COVO_NA) {
            v = 0.0f;
#pragma unroll
            for (int i = 0; i < MG_SLICES; ++i) v += sv[i][tid];
            rec[2 + tid] = v;
        }
        for (int c2 = tid; c2 < RD_COV_FLOATS; c2 += MG_THREADS) {
            float v2 = 0.0f;
#pragma unroll
            for (int i = 0; i < MG_SLICES; ++i) v2 += sv2[i][c2];
            rec[COVO_PARTIAL_FLOATS + c2] = v2;
        }
        if (tid == 0) {
            rec[0] = m;
            rec[1] = s;
        }
        return;
    }
    const float inv_s = 1.0f / s;
    if (tid < COVO_NA) {
        v = 0.0f;
#pragma unroll
        for (int i = 0; i < MG_SLICES; ++i) v += sv[i][tid];
        const float wmean = v * inv_s, mu = a_mean_old[tid];
        const float mean_new = wmean * gamma_mean + mu * (1.0f - gamma_mean);  // mppi.py:112-117
        smean[tid] = mean_new;
        sm1[tid] = wmean - mu;
        se[tid] = mean_new - mu;
    }
    for (int c2 = tid; c2 < RD_COV_FLOATS; c2 += MG_THREADS) {
        float v2 = 0.0f;
#pragma unroll
        for (int i = 0; i < MG_SLICES; ++i) v2 += sv2[i][c2];
        s2[c2] = v2 * inv_s;
    }
    __syncthreads();
    if (tid < COVO_NA) a_mean_out[tid] = smean[tid];
    if (tid < COVO_H * 16) {
        const int t = tid >> 4, i = (tid >> 2) & 3, j = tid & 3;
        const float c = s2[10 * t + cov_pair(i, j)] - sm1[4 * t + i] * se[4 * t + j] - se[4 * t + i] * sm1[4 * t + j] +
                        se[4 * t + i] * se[4 * t + j];
        a_cov_out[tid] = c * gamma_sigma + a_cov_old[tid] * (1.0f - gamma_sigma);  // mppi.py:119-125 (in place is fine: own element)
    }
}

// Merges G records {m, s, v[128]} with 1024 threads = 8 record-slices x 128 columns (softmax_merge.hpp: the body is shared
// with the launches that finish their own update).
// FINAL: a_mean_out = gamma * v/s + (1-gamma) * a_mean_old (covo.py:270-275); otherwise writes the
// merged record to out.  Fixed summation order -> bit-reproducible.
// stride: floats between consecutive records (COVO_PARTIAL_FLOATS, or COVO_RANK_RECORD_FLOATS for the all-gathered rank records
// that also carry the position sums)
template <bool FINAL>
__global__ __launch_bounds__(MG_THREADS) void merge_kernel(const float *__restrict__ partials, int G, float inv_lam,
                                                           const float *__restrict__ a_mean_old, float gamma_mean,
                                                           float *__restrict__ out, int stride)
{
    __shared__ MergeLds lds;
    {   // blockIdx.x (env-batched step): instance x merges its own G records into its own mean
        const size_t x = blockIdx.x;
        partials += x * G * stride;
        if (FINAL) a_mean_old += x * COVO_NA;
        out += x * (FINAL ? COVO_NA : COVO_PARTIAL_FLOATS);
    }
    merge_body<MG_THREADS, FINAL, false>(partials, G, inv_lam, a_mean_old, gamma_mean, out, stride, lds);
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
    hipLaunchKernelGGL(softmax_partial_kernel<false>, dim3(grid, batch), dim3(RD_BLOCK), 0, s, cost,
                       reinterpret_cast<const float4 *>(a), N, blockmin, n_blockmin, inv_lam, partials_ws, (const float4 *)nullptr);
    if (a_mean_out != nullptr)
        hipLaunchKernelGGL(merge_kernel<true>, dim3(batch), dim3(MG_THREADS), 0, s, partials_ws, grid, inv_lam, a_mean_old,
                           gamma_mean, a_mean_out, COVO_PARTIAL_FLOATS);
    else
        hipLaunchKernelGGL(merge_kernel<false>, dim3(batch), dim3(MG_THREADS), 0, s, partials_ws, grid, inv_lam,
                           (const float *)nullptr, 1.0f, partial_out, COVO_PARTIAL_FLOATS);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}

// MPPI's update with covariance adaptation (mppi.py:109-125): stage 1 with second moments, then merge_cov_kernel; single shard
size_t softmax_cov_workspace_floats(int max_blocks) { return (size_t)max_blocks * RD_COV_RECORD_FLOATS; }
int launch_softmax_update_cov(covo_ctx *h, const float *cost, const float *a, int N, const float *blockmin, int n_blockmin,
                              const float *a_mean_old, float gamma_mean, const float *a_cov_old, float gamma_sigma,
                              float *a_mean_out, float *a_cov_out, hipStream_t s)
{
    const float inv_lam = 1.0f / h->cfg.lam;
    if (blockmin == nullptr) {
        n_blockmin = (N + 63) / 64;
        hipLaunchKernelGGL(groupmin_kernel, dim3((N + 255) / 256), dim3(256), 0, s, cost, N, h->ws_blockmin);
        blockmin = h->ws_blockmin;
    }
    const int ngroups = (N + 63) / 64;
    int grid = (ngroups + RD_WAVES - 1) / RD_WAVES;
    if (grid > h->max_red_blocks) grid = h->max_red_blocks;
    hipLaunchKernelGGL(softmax_partial_kernel<true>, dim3(grid, 1), dim3(RD_BLOCK), 0, s, cost, reinterpret_cast<const float4 *>(a), N,
                       blockmin, n_blockmin, inv_lam, h->ws_partials_cov, reinterpret_cast<const float4 *>(a_mean_old));
    if (a_cov_out != nullptr)
        hipLaunchKernelGGL(merge_cov_kernel<true>, dim3(1), dim3(MG_THREADS), 0, s, h->ws_partials_cov, grid, inv_lam, a_mean_old,
                           gamma_mean, a_cov_old, gamma_sigma, a_mean_out, a_cov_out, RD_COV_RECORD_FLOATS);
    else  // a sample-sharded rank: a_mean_out = this rank's record {m, s, v, pad, S2} (launch_softmax_reduce_cov)
        hipLaunchKernelGGL(merge_cov_kernel<false>, dim3(1), dim3(MG_THREADS), 0, s, h->ws_partials_cov, grid, inv_lam, a_mean_old,
                           1.0f, (const float *)nullptr, 0.0f, a_mean_out, (float *)nullptr, RD_COV_RECORD_FLOATS);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}

// stage 1 with second moments + local merge -> this rank's record {m, s, v[128], pad[2], S2[320]} (unnormalised): the first
// COVO_PARTIAL_FLOATS + 320 floats of a COVO_RANK_RECORD_COV_FLOATS rank record
int launch_softmax_reduce_cov(covo_ctx *h, const float *cost, const float *a, int N, const float *blockmin, int n_blockmin,
                              const float *a_mean_old, float *record_out, hipStream_t s)
{
    return launch_softmax_update_cov(h, cost, a, N, blockmin, n_blockmin, a_mean_old, 1.0f, nullptr, 0.0f, record_out, nullptr, s);
}

// the G all-gathered rank records (stride floats apart) -> new mean and adapted covariances, identically on every rank
int launch_merge_cov(const float *records, int G, int stride, float lam, const float *a_mean_old, float gamma_mean,
                     const float *a_cov_old, float gamma_sigma, float *a_mean_out, float *a_cov_out, hipStream_t s)
{
    if (G > MG_MAXG) { covo_set_error("covo_merge_ranks_cov: G=%d > %d", G, MG_MAXG); return COVO_E_BADARG; }
    hipLaunchKernelGGL(merge_cov_kernel<true>, dim3(1), dim3(MG_THREADS), 0, s, records, G, 1.0f / lam, a_mean_old, gamma_mean,
                       a_cov_old, gamma_sigma, a_mean_out, a_cov_out, stride);
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
