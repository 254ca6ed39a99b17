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
//   * L is staged once per workgroup into a padded LDS image (66 KiB) and its A-fragments are read
//     with conflict-free ds_read_b32 at the point of use; <= 256 registers per lane keep TWO waves
//     resident per SIMD, which is what hides everything that is not an MFMA: a single wave issues
//     one non-MFMA instruction per ~5 cycles in order, so at 1 wave/SIMD the epilogue, the swaps
//     and the epsilon generation all serialise with the matrix pipe (measured: 77 -> see DESIGN.md).
//   * B fragments come straight from global memory: lanes (j, kh) of a wave load float4 chunk
//     2c+kh of sample row j (the lane pair consumes the whole 512-B row, full 128-B lines), and
//     two v_permlane32_swap per chunk pair put elements {k, k+1} on lanes {j, j+32} as the
//     32x32x2 B operand wants them.
//   * the D layout (col = lane&31 -> sample, rows 8g+4h+i -> (t, d)) makes every lane own whole
//     float4 actions, so the epilogue (+mu, clip) stores 512-B contiguous runs of the stripe
//     layout a[H][N][4] the rollout kernel reads.
//   * launch shapes (round 3): 512-thread workgroups once the launch fills the chip (>= 2 048 tiles: the factor is staged once per
//     CU), 256-thread ones below, and for <= 512 tiles every tile is shared by two waves (row tiles {0, 3} / {1, 2}: 80 MFMAs
//     each) -- noise_gemm_block_threads / noise_gemm_split below; all shapes give the same bits.
// fp32 MFMA roofline: 2*128*128 flop per sample (dense-equivalent), 157.3 TFLOP/s peak.
#include "covo_common.hpp"
#include <cstdlib>
#include <cstring>
#include "eps_tiles.hpp"

// scripts/probe/gemm_probe.hip compiles this file with GEMM_PROBE: wave 0 of every workgroup leaves s_memrealtime
// stamps (100 MHz) at the phase boundaries of its first tile.  Compiled out of the library.
#ifdef GEMM_PROBE
__device__ unsigned long long *g_ng_probe;
#define NG_STAMP(i)                                                                                         \
    if (threadIdx.x == 0 && g_ng_probe) g_ng_probe[8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + (i)] = wall_clock64();
#else
#define NG_STAMP(i)
#endif
#include "noise_gemm_body.hpp"  // (after NG_STAMP: the factor staging carries two of the probe's stamps)

// PHILOX = false: epsilon is read from `eps` (TILED: in the tile order of eps_tiles.hpp, else row-major (N, 128));
// true: drawn in registers from (k0, k1, sample_offset + n).
// NG_BLOCK threads per workgroup: 512 = one workgroup per CU, the factor staged into LDS once per CU -- the choice once a launch
// fills the chip (>= 2 048 tiles: 20.2 -> 17.8 us at N = 65 536); 256 = two per CU: smaller launches spread over twice as many
// CUs (N = 8 192: 64 workgroups with one wave per SIMD instead of 32 with two; covo-offline 36.2k against 29.5k steps/s).
// noise_gemm_block_threads() below decides; the rollout's XCD-affine mapping follows it.
// SPLIT (launches of <= 512 tiles, i.e. N <= 16 384: fewer tiles than half the chip's SIMDs): a tile is shared by TWO waves -- row
// tiles {0, 3} and {1, 2}, 80 of the 160 MFMAs each -- so that twice as many SIMDs work and a wave's serial MFMA time halves
// (N = 8 192: 256 tiles on 1 024 SIMDs).  Same dot products, same order: bit-identical to the unsplit kernel.
template <bool PHILOX, bool TILED = false, int NG_BLOCK = 256, bool SPLIT = false>
__global__ __launch_bounds__(NG_BLOCK, 512 / NG_BLOCK) void noise_gemm_kernel(const float *__restrict__ L, const float *__restrict__ mu,
                                                              const float *__restrict__ eps, uint32_t k0, uint32_t k1,
                                                              int64_t sample_offset, int N, int ntiles,
                                                              float4 *__restrict__ a_out, const uint32_t *__restrict__ dyn,
                                                              const float *__restrict__ state_for_time, int n_table,
                                                              const CovDeferred cov, int nanp)
{
    // dyn (nullable): {key0, key1} in device memory -- lets a captured graph see a fresh key every replay.
    // state_for_time (nullable): L is a table [n_table][128][128]; use row state.time (covo.py:107-108, clamped
    // like a JAX gather).
    // blockIdx.y (env-batched step): instance y has its own factor, mean, key block and action stripes, all dense
    {
        const size_t y = blockIdx.y;
        L += y * (COVO_NA * COVO_NA);
        mu += y * COVO_NA;
        a_out += y * ((size_t)COVO_H * N);
        if (dyn != nullptr) dyn += y * 12;
        if (TILED) eps += y * ((size_t)ntiles * 16 * 64 * 4);  // the instance's tiles (eps_tiles.hpp)
    }
    NG_STAMP(0);
    if (dyn != nullptr) { k0 = dyn[0]; k1 = dyn[1]; }
    if (state_for_time != nullptr) {
        int t = __float_as_int(state_for_time[ST_TIME]);
        t = t < 0 ? 0 : (t > n_table - 1 ? n_table - 1 : t);
        L += (size_t)t * COVO_NA * COVO_NA;
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ls = smem;                       // [128][NG_LDA]
    float *mus = smem + COVO_NA * NG_LDA;   // [128]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kh = lane >> 5;

    const int wave_global = blockIdx.x * (NG_BLOCK / 64) + wave;
    const int wave_stride = gridDim.x * (NG_BLOCK / 64);
    auto rowptr = [&](int t) {
        int row = t * 32 + j;
        row = row < N ? row : N - 1;
        return reinterpret_cast<const float4 *>(eps + (size_t)row * COVO_NA);
    };
    int item = wave_global;  // work item: a tile, or (SPLIT) half of one
    const int nitems = SPLIT ? 2 * ntiles : ntiles;
    auto tile_id = [&](int it) { return SPLIT ? (it >> 1) : it; };
    // the first tile's epsilon rows are requested before anything else: their HBM latency hides
    // behind the staging of L below
    auto tile_of = [&](int t) {
        if (PHILOX) {
            int row = t * 32 + j;
            row = row < N ? row : N - 1;
            return gen_tile((uint64_t)(sample_offset + row), kh, k0, k1);
        }
        if (TILED) return load_tile_tiled(reinterpret_cast<const float4 *>(eps) + (size_t)t * 16 * 64, lane);
        return load_tile(rowptr(t), kh);
    };
    // PHILOX: only k-group 0 of the first tile is drawn up front; every later group is drawn right behind the MFMAs
    // of the group before it (program order), so it executes while the matrix pipe works through those MFMAs --
    // this wave's own and those of the other wave resident on the SIMD.
    auto id_of = [&](int t) {
        int row = t * 32 + j;
        row = row < N ? row : N - 1;
        return (uint64_t)(sample_offset + row);
    };
    // STREAM (in-kernel Philox, or the tile-ordered image): one k-group is fetched at a time, one group ahead of its MFMAs
    constexpr bool STREAM = PHILOX || TILED;
    auto fetch_group = [&](int t, int g) {
        if (PHILOX) return gen_group(id_of(t), g, kh, k0, k1);
        BGroup b;
        const float4 *base = reinterpret_cast<const float4 *>(eps) + ((size_t)t * 16 + 4 * g) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) b.c[i] = base[i * 64];
        return b;
    };
    BTile cur;
    if (STREAM) cur.g[0] = fetch_group(item < nitems ? tile_id(item) : 0, 0);
    else cur = tile_of(item < nitems ? tile_id(item) : 0);
    __builtin_amdgcn_sched_barrier(0);
    // a_cov = cz sym(Z), left over by the Sigma chain's single-workgroup finalize launch (CovDeferred): the first 64
    // workgroups take 256 elements each; the loads ride with the epsilon request above, the stores leave after the staging of L
    double cov_z = 0.0, cov_zt = 0.0, cov_cz = 0.0, cov_zc = 0.0, cov_ur = 0.0, cov_uc = 0.0;
    const bool cov_on = cov.out != nullptr && blockIdx.y == 0 && blockIdx.x < COVO_NA * COVO_NA / NG_BLOCK;
    const int cov_zb = cov_on ? ((*cov.zbuf != 0.0) ? 1 : 0) : 0;
    if (cov_on) {
        cov_cz = *cov.cz;
        cov_zc = *cov.zcoef;  // deflated chain: Z = Z~ + zc u u^T (sigma_ns.hip)
        if (gridDim.x >= COVO_NA * COVO_NA / NG_BLOCK) {
            const int e = blockIdx.x * NG_BLOCK + tid;
            cov_z = cov.Z[cov_zb][e];
            cov_zt = cov.Zt[cov_zb][e];
            if (cov_zc != 0.0) {
                cov_ur = cov_zc * cov.u[e / COVO_NA];
                cov_uc = cov.u[e % COVO_NA];
            }
        }
    }

    // ---- stage L (masked to its lower triangle) and mu.  All of a thread's loads are issued before the first LDS
    // write (a rolled load -> write loop pays one L2 round trip per trip: 3.8 us of a 17 us launch, scripts/probe/
    // gemm_probe.hip); chunks right of the diagonal block are never read by the MFMA loop (row tile rt stops at
    // k < 32 (rt + 1)) and are neither loaded nor written, chunks right of the diagonal inside it are written as zeros.
    ng_stage_factor<NG_BLOCK>(L, Ls, tid);
    if (tid < COVO_NA) mus[tid] = mu[tid];
    if (cov_on) {
        if (gridDim.x >= COVO_NA * COVO_NA / NG_BLOCK) {
            // covo.py:132 symmetrise (+ the deflated eigenpair, the same expression as ns_finalize_kernel's); a_cov is fp32
            double v = 0.5 * (cov_z + cov_zt);
            if (cov_zc != 0.0) v = fma(cov_ur, cov_uc, v);
            cov.out[blockIdx.x * NG_BLOCK + tid] = (float)(cov_cz * v);
        } else {  // small launches (N < 16 384): the workgroups there are stride over the matrix
            for (int e = blockIdx.x * NG_BLOCK + tid; e < COVO_NA * COVO_NA; e += gridDim.x * NG_BLOCK) {
                double v = 0.5 * (cov.Z[cov_zb][e] + cov.Zt[cov_zb][e]);
                if (cov_zc != 0.0) v = fma(cov_zc * cov.u[e / COVO_NA], cov.u[e % COVO_NA], v);
                cov.out[e] = (float)(cov_cz * v);
            }
        }
    }
    __syncthreads();

    const float *__restrict__ La = Ls + j * NG_LDA + kh;  // this lane's row / k-parity of every A fragment
    NG_STAMP(1);

    if (item >= nitems) return;

    for (; item < nitems; item += wave_stride) {
        const int tile = tile_id(item);
        f32x16 acc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rt][e] = 0.0f;

        const int next_item = item + wave_stride;
        const bool has_next = next_item < nitems;
        const int next_tile = has_next ? tile_id(next_item) : tile;
        BTile nxt;
        if (!STREAM) {
            // whole next tile in flight while this one is multiplied (10 240 MFMA cycles of cover)
            nxt = tile_of(next_tile);
            __builtin_amdgcn_sched_barrier(0);
        }
        // L is lower triangular: row tile rt (actions of steps 8 rt .. 8 rt + 7) is complete after k-group rt, so its
        // epilogue (+ mu, clip, stripe-ordered float4 stores) goes out right there and its 8 KiB per wave drain to HBM
        // under the remaining MFMAs.  (With all four row tiles stored at the end, a launch in which every wave owns
        // one tile computes for ~11 us and then waits for 33.5 MB of stores to drain.)
        const int n = tile * 32 + j;
        auto store_rt = [&](int rt) {
            if (n < N) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int t = 8 * rt + 2 * g + kh;
                    const float4 m4 = *reinterpret_cast<const float4 *>(mus + 4 * t);
                    float4 v;
                    if (nanp) {  // COVO_FLAG_PROPAGATE_NAN (wave-uniform): a NaN sample stays NaN, as under jnp.clip (covo.py:224)
                        v.x = qm::clip11_nan_(m4.x + acc[rt][4 * g + 0]);
                        v.y = qm::clip11_nan_(m4.y + acc[rt][4 * g + 1]);
                        v.z = qm::clip11_nan_(m4.z + acc[rt][4 * g + 2]);
                        v.w = qm::clip11_nan_(m4.w + acc[rt][4 * g + 3]);
                    } else {
                        v.x = qm::clip11_(m4.x + acc[rt][4 * g + 0]);
                        v.y = qm::clip11_(m4.y + acc[rt][4 * g + 1]);
                        v.z = qm::clip11_(m4.z + acc[rt][4 * g + 2]);
                        v.w = qm::clip11_(m4.w + acc[rt][4 * g + 3]);
                    }
                    // -DNG_STORE_WT / -DNG_STORE_NT (write-through / nontemporal stores, so that the 33.5 MB do not wait dirty in
                    // the L2s for the end-of-kernel write-back): 24.6 -> 21.3 us in the isolated replay of the in-kernel-Philox
                    // variant, but bench.py on every config moves by less than its run-to-run spread (4 791 / 4 824 / 4 786
                    // covo-online, 22 390 / 22 845 / 22 140 covo-offline): the rollout then misses the L2 on every stripe.  Plain stores stay.
#ifdef NG_STORE_WT
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    const f4v vv = {v.x, v.y, v.z, v.w};
                    // (s_nop 1: the > 64-bit-store data hazard the compiler cannot see inside inline asm, eps_tiles.hpp)
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(a_out + (size_t)t * N + n), "v"(vv) : "memory");
#elif defined(NG_STORE_NT)
                    __builtin_nontemporal_store(v.x, &a_out[(size_t)t * N + n].x);
                    __builtin_nontemporal_store(v.y, &a_out[(size_t)t * N + n].y);
                    __builtin_nontemporal_store(v.z, &a_out[(size_t)t * N + n].z);
                    __builtin_nontemporal_store(v.w, &a_out[(size_t)t * N + n].w);
#else
                    a_out[(size_t)t * N + n] = v;
#endif
                }
            }
        };
        // the sequence for the row-tile set M of this work item (15: the whole tile); k-group g is fetched / multiplied only if a
        // row tile >= g is in the set, row tile g stored right after k-group g if it is in the set
        auto seq = [&](auto mtag) {
            constexpr int M = decltype(mtag)::value;
            if (TILED && (M >> 1)) cur.g[1] = fetch_group(tile, 1);  // loads: requested BEFORE the MFMAs that hide them
            mfma_group<0, M>(La, cur.g[0], acc);
            NG_STAMP(2);
            if (M & 1) store_rt(0);
            if (PHILOX && (M >> 1)) cur.g[1] = fetch_group(tile, 1);  // draws: issued BEHIND the MFMAs they overlap with
            if (TILED && (M >> 2)) cur.g[2] = fetch_group(tile, 2);
            mfma_group<1, M>(La, cur.g[1], acc);
            NG_STAMP(3);
            if (M & 2) store_rt(1);
            if (PHILOX && (M >> 2)) cur.g[2] = fetch_group(tile, 2);
            if (TILED && (M >> 3)) cur.g[3] = fetch_group(tile, 3);
            mfma_group<2, M>(La, cur.g[2], acc);
            NG_STAMP(4);
            if (M & 4) store_rt(2);
            if (PHILOX && (M >> 3)) cur.g[3] = fetch_group(tile, 3);
            if (TILED && has_next) cur.g[0] = fetch_group(next_tile, 0);
            mfma_group<3, M>(La, cur.g[3], acc);
            NG_STAMP(5);
            if (M & 8) store_rt(3);
            NG_STAMP(6);
            if (PHILOX && has_next) cur.g[0] = fetch_group(next_tile, 0);
        };
        if (!SPLIT) seq(RtMask<15>());
        else if (item & 1) seq(RtMask<6>());
        else seq(RtMask<9>());
        if (!STREAM) cur = nxt;
    }
}

// MPPI: a[t] = clip(mu[t] + Ls[t] eps[t]) with 4x4 lower factors; eps [N][H][4] -> a [H][N][4].
// HBM-bound streaming kernel (1 KiB per sample); fmaf chains in ascending k like the oracle.
// PHILOX (the product path: nothing is read): workgroup row y = step t, consecutive lanes = consecutive samples -- a wave stores
// 1 KiB contiguous runs of stripe t.  (Round 3; the first mapping, consecutive lanes = consecutive t of one sample, suits the eps
// READ of the other variant but scattered the stores as 64 partial lines per wave: 19.2 us at N = 65 536.)
template <bool PHILOX>
__global__ __launch_bounds__(256) void noise_blockdiag_kernel(const float *__restrict__ Ls, const float *__restrict__ mu,
                                                              const float4 *__restrict__ eps, uint32_t k0, uint32_t k1,
                                                              int64_t sample_offset, int N, float4 *__restrict__ a_out,
                                                              const uint32_t *__restrict__ dyn, int nanp)
{
    if (dyn != nullptr) { k0 = dyn[0]; k1 = dyn[1]; }
    int t;
    size_t n;
    float Lt[16], mt[4];
    if (PHILOX) {
        t = blockIdx.y;
        n = (size_t)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
        for (int i = 0; i < 16; ++i) Lt[i] = Ls[t * 16 + i];  // (uniform: scalar loads)
#pragma unroll
        for (int i = 0; i < 4; ++i) mt[i] = mu[t * 4 + i];
        if (n >= (size_t)N) return;
    } else {
        // thread -> (sample, t): consecutive threads take consecutive t of one sample (coalesced eps reads)
        const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
        if (gid >= (size_t)N * COVO_H) return;
        t = (int)(gid % COVO_H);
        n = gid / COVO_H;
#pragma unroll
        for (int i = 0; i < 16; ++i) Lt[i] = Ls[t * 16 + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) mt[i] = mu[t * 4 + i];
    }
    const float4 e = PHILOX ? rngd::normal4((uint32_t)t, (uint64_t)(sample_offset + (int64_t)n), k0, k1) : eps[n * COVO_H + t];
    const float ev[4] = {e.x, e.y, e.z, e.w};
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = fmaf((k <= i) ? Lt[i * 4 + k] : 0.0f, ev[k], acc);
        o[i] = nanp ? qm::clip11_nan_(mt[i] + acc) : qm::clip11_(mt[i] + acc);  // mppi.py:66 (COVO_FLAG_PROPAGATE_NAN)
    }
    a_out[(size_t)t * N + n] = make_float4(o[0], o[1], o[2], o[3]);
}

// (batched launches with >= 2 048 tiles in all: 8-wave workgroups and at most 256 of them, every wave strides over its instance's
// tiles -- the factor is staged 256 times instead of 1 024: 43.1 -> 39.8 us at 32 instances x N = 4 096; the launch stays bound by
// its 17.8 us of MFMAs + the in-kernel Philox, which the matrix pipe does not hide)
int noise_gemm_block_threads(int N, int batch) { return ((long long)((N + 31) / 32) * batch >= 2048) ? 512 : 256; }
// launches of <= 512 tiles share every tile between two waves (the kernel's SPLIT)
static bool noise_gemm_split(int N, int batch) { return (long long)((N + 31) / 32) * batch <= 512; }
// 64-sample groups per workgroup (the rollout's XCD-affine mapping, rollout.hip)
int noise_gemm_groups_per_workgroup(int N, int batch)
{
    return noise_gemm_split(N, batch) ? 1 : noise_gemm_block_threads(N, batch) / 128;
}

int launch_noise_gemm(const float *L, const float *mu, const float *eps, uint32_t k0, uint32_t k1, int64_t sample_offset,
                      int N, float *a, hipStream_t s, const uint32_t *dyn, const float *state_for_time, int n_table, int batch,
                      bool eps_tiled, const CovDeferred *cov, bool propagate_nan)
{
    CovDeferred cv;
    std::memset(&cv, 0, sizeof(cv));
    if (cov != nullptr) cv = *cov;
    const int ntiles = (N + 31) / 32;
    const int nanp = propagate_nan ? 1 : 0;
    const int block = noise_gemm_block_threads(N, batch);
    const bool split = noise_gemm_split(N, batch);
    const int waves_per_block = block / 64;
    int grid = ((split ? 2 * ntiles : ntiles) + waves_per_block - 1) / waves_per_block;
    if (grid > 2048 / waves_per_block) grid = 2048 / waves_per_block;  // persistent: 2 waves/SIMD chip-wide, waves stride over tiles
    if (batch > 1 && block == 512 && (long long)grid * batch > 256) grid = (256 + batch - 1) / batch;  // one 8-wave workgroup per CU
    const size_t lds = (size_t)(COVO_NA * NG_LDA + COVO_NA) * sizeof(float);
    static unsigned long long attr_devices = 0;  // (per device: covo_first_on_device)
    if (covo_first_on_device(attr_devices)) {
#define NG_ATTR(...) COVO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(noise_gemm_kernel<__VA_ARGS__>), \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
        NG_ATTR(false, false, 256); NG_ATTR(true, false, 256); NG_ATTR(false, true, 256);
        NG_ATTR(false, false, 512); NG_ATTR(true, false, 512); NG_ATTR(false, true, 512);
        NG_ATTR(false, false, 256, true); NG_ATTR(true, false, 256, true); NG_ATTR(false, true, 256, true);
#undef NG_ATTR
    }
#define NG_GO(...)                                                                                                             \
    do {                                                                                                                         \
        if (eps != nullptr && eps_tiled)                                                                                         \
            hipLaunchKernelGGL((noise_gemm_kernel<false, true, __VA_ARGS__>), dim3(grid, batch), dim3(block), lds, s, L, mu, eps, 0u, 0u, (int64_t)0, \
                               N, ntiles, reinterpret_cast<float4 *>(a), (const uint32_t *)nullptr, state_for_time, n_table, cv, nanp); \
        else if (eps != nullptr)                                                                                                 \
            hipLaunchKernelGGL((noise_gemm_kernel<false, false, __VA_ARGS__>), dim3(grid), dim3(block), lds, s, L, mu, eps, 0u, 0u,          \
                               (int64_t)0, N, ntiles, reinterpret_cast<float4 *>(a), (const uint32_t *)nullptr, state_for_time,  \
                               n_table, cv, nanp);                                                                               \
        else                                                                                                                     \
            hipLaunchKernelGGL((noise_gemm_kernel<true, false, __VA_ARGS__>), dim3(grid, batch), dim3(block), lds, s, L, mu,                 \
                               (const float *)nullptr, k0, k1, sample_offset, N, ntiles, reinterpret_cast<float4 *>(a), dyn,     \
                               state_for_time, n_table, cv, nanp);                                                               \
    } while (0)
    if (block == 512) NG_GO(512);
    else if (split) NG_GO(256, true);
    else NG_GO(256);
#undef NG_GO
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_noise_blockdiag(const float *Ls, const float *mu, const float *eps, uint32_t k0, uint32_t k1,
                           int64_t sample_offset, int N, float *a, hipStream_t s, const uint32_t *dyn, bool propagate_nan)
{
    const int nanp = propagate_nan ? 1 : 0;
    const size_t total = (size_t)N * COVO_H;
    const int grid = (int)((total + 255) / 256);
    if (eps != nullptr)
        hipLaunchKernelGGL(noise_blockdiag_kernel<false>, dim3(grid), dim3(256), 0, s, Ls, mu,
                           reinterpret_cast<const float4 *>(eps), 0u, 0u, (int64_t)0, N, reinterpret_cast<float4 *>(a),
                           (const uint32_t *)nullptr, nanp);
    else
        hipLaunchKernelGGL(noise_blockdiag_kernel<true>, dim3((N + 255) / 256, COVO_H), dim3(256), 0, s, Ls, mu,
                           (const float4 *)nullptr, k0, k1, sample_offset, N, reinterpret_cast<float4 *>(a), dyn, nanp);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}
