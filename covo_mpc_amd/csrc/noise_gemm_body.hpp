// noise_gemm_body.hpp -- the device pieces of the correlated-noise draw a = clip(mu + L eps) on v_mfma_f32_32x32x2_f32, shared by
// noise_gemm_kernel (noise_gemm.hip) and the fused small step (step_small.hip) so that both form every dot product with the
// same instructions in the same order (bit-identical actions).  See noise_gemm.hip for the design notes.
#pragma once
#include "covo_common.hpp"
#include "rng_device.hpp"

#ifndef NG_STAMP
#define NG_STAMP(i)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NG_LDA = COVO_NA + 1;  // padded leading dimension of the LDS image of L


struct BGroup {
    float4 c[4];  // this lane's chunks (2*(4g+i) + kh) of its sample row, i = 0..3
};
struct BTile {
    BGroup g[4];  // the lane's half of its 512-B sample row: 16 x float4 (64 VGPRs)
};

__device__ __forceinline__ BTile load_tile(const float4 *__restrict__ row, int kh)
{
    BTile t;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) t.g[g].c[i] = row[2 * (4 * g + i) + kh];
    return t;
}

// the same 16 chunks from the tile-ordered image drawn ahead of the GEMM (eps_tiles.hpp): 16 contiguous 1-KiB wave loads
__device__ __forceinline__ BTile load_tile_tiled(const float4 *__restrict__ tile_base, int lane)
{
    BTile t;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) t.g[g].c[i] = tile_base[(4 * g + i) * 64 + lane];
    return t;
}

// the same 16 chunks drawn in place: chunk index = Philox counter word 0 (rng_device.hpp), so the
// values equal randn_kernel's for the same (key, global sample id)
__device__ __forceinline__ BTile gen_tile(uint64_t id, int kh, uint32_t k0, uint32_t k1)
{
    BTile t;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) t.g[g].c[i] = rngd::normal4((uint32_t)(2 * (4 * g + i) + kh), id, k0, k1);
    return t;
}

// One k-group (32 k's = 16 k-steps) of the tile.  A fragments A[i = lane&31][k = 2 ks + (lane>>5)] are
// read from the padded LDS image of L right where they are used (conflict-free ds_read_b32; two
// resident waves per SIMD hide their latency), so the kernel fits 2 waves/SIMD and one wave's
// epilogue / epsilon generation overlaps the other's MFMAs.
__device__ __forceinline__ BGroup gen_group(uint64_t id, int g, int kh, uint32_t k0, uint32_t k1)
{
    BGroup b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        b.c[i] = rngd::normal4((uint32_t)(2 * (4 * g + i) + kh), id, k0, k1);
        __builtin_amdgcn_sched_barrier(0);  // one Philox live at a time: keeps the kernel under 256 VGPRs
    }
    return b;
}

// A fragments of one float4 chunk of k's (4 k-steps x the row tiles rt >= G of the wave's set MASK), read from the LDS image of L.
// MASK: which row tiles (32 actions each) this wave multiplies -- 15: all (a whole tile per wave); 9 / 6: {0, 3} / {1, 2}, the two
// halves of a tile with 80 of its 160 MFMAs each (small launches: a tile split over two waves, see the kernel)
template <int G>
struct AFrag {
    float v[4][4];  // [q][rt] (entries outside the set are never touched)
};
template <int G, int MASK>
__device__ __forceinline__ AFrag<G> load_afrag(const float *__restrict__ La, int i)
{
    AFrag<G> f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ks = 16 * G + 4 * i + q;  // k-step (k0 = 2 ks)
#pragma unroll
        for (int rt = G; rt < 4; ++rt)
            if ((MASK >> rt) & 1) f.v[q][rt] = La[32 * rt * NG_LDA + 2 * ks];
    }
    return f;
}

// One k-group (32 k's) of the tile.  The A fragments of chunk i+1 are requested from LDS BEFORE the MFMAs of chunk
// i are issued (a ds_read -> s_waitcnt -> 2 dependent MFMAs sequence, as hipcc schedules the naive loop, leaves the
// matrix pipe idle for an LDS round trip per pair: 26.8 us; with the fragments one chunk ahead the MFMAs of a chunk
// go out back to back on four independent accumulators).
template <int G, int MASK = 15>
__device__ __forceinline__ void mfma_group(const float *__restrict__ La, BGroup b, f32x16 (&acc)[4])
{
    if ((MASK >> G) == 0) return;  // no row tile of the set reaches this k-group (L is lower triangular)
    AFrag<G> cur = load_afrag<G, MASK>(La, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        AFrag<G> nxt = cur;
        if (i < 3) nxt = load_afrag<G, MASK>(La, i + 1);
        float x = b.c[i].x, y = b.c[i].y, z = b.c[i].z, w = b.c[i].w;
        // lanes 32-63 of vdst <-> lanes 0-31 of src
        auto r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(z), __float_as_uint(w), false, false);
        const float bk0 = __uint_as_float(r0[0]);  // k = 8c+0 | 8c+1
        const float bk4 = __uint_as_float(r0[1]);  // k = 8c+4 | 8c+5
        const float bk2 = __uint_as_float(r1[0]);  // k = 8c+2 | 8c+3
        const float bk6 = __uint_as_float(r1[1]);  // k = 8c+6 | 8c+7
        const float bb[4] = {bk0, bk2, bk4, bk6};
        __builtin_amdgcn_sched_barrier(0);  // keep the next chunk's LDS reads above this chunk's MFMAs
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int rt = G; rt < 4; ++rt)
                if ((MASK >> rt) & 1) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.v[q][rt], bb[q], acc[rt], 0, 0, 0);
        }
        cur = nxt;
    }
}

// Half of mfma_group<G, MASK>: the chunks {2 HALF, 2 HALF + 1} of k-group G, i.e. the 16 columns of ONE Cholesky panel (2 G + HALF)
// -- the streamed finalize launch multiplies panel by panel as the factor arrives.  The two halves in order are mfma_group: the
// same fragments, the same MFMAs in the same order on the same accumulators.
template <int G, int HALF, int MASK = 15>
__device__ __forceinline__ void mfma_half(const float *__restrict__ La, const BGroup &b, f32x16 (&acc)[4])
{
    if ((MASK >> G) == 0) return;
    AFrag<G> cur = load_afrag<G, MASK>(La, 2 * HALF);
#pragma unroll
    for (int i = 2 * HALF; i < 2 * HALF + 2; ++i) {
        AFrag<G> nxt = cur;
        if (i < 2 * HALF + 1) nxt = load_afrag<G, MASK>(La, i + 1);
        float x = b.c[i].x, y = b.c[i].y, z = b.c[i].z, w = b.c[i].w;
        auto r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(z), __float_as_uint(w), false, false);
        const float bb[4] = {__uint_as_float(r0[0]), __uint_as_float(r1[0]), __uint_as_float(r0[1]), __uint_as_float(r1[1])};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int rt = G; rt < 4; ++rt)
                if ((MASK >> rt) & 1) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.v[q][rt], bb[q], acc[rt], 0, 0, 0);
        }
        cur = nxt;
    }
}

// a row-tile set as a type (generic lambdas take it as a tag)
template <int M>
struct RtMask {
    static constexpr int value = M;
};

// Stage L (masked to its lower triangle) into the padded LDS image.  All of a thread's loads are issued before the first LDS
// write (a rolled load -> write loop pays one L2 round trip per trip: 3.8 us of a 17 us launch, scripts/probe/
// gemm_probe.hip); chunks right of the diagonal block are never read by the MFMA loop (row tile rt stops at
// k < 32 (rt + 1)) and are neither loaded nor written, chunks right of the diagonal inside it are written as zeros.
// Called by all NG_BLOCK threads of the workgroup; the caller synchronises before the first fragment read.
template <int NG_BLOCK>
__device__ __forceinline__ void ng_stage_factor(const float *__restrict__ L, float *__restrict__ Ls, const int tid)
{
    constexpr int TRIPS = COVO_NA * COVO_NA / 4 / NG_BLOCK;  // 16
    const int k4 = (tid & 31) * 4, i0 = tid >> 5;            // trip `it` handles row i0 + 8 it, columns k4 .. k4 + 3
    float4 v[TRIPS];
#pragma unroll
    for (int it = 0; it < TRIPS; ++it) {
        const int i = i0 + (NG_BLOCK / 32) * it;
        v[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (k4 <= i) v[it] = reinterpret_cast<const float4 *>(L)[tid + NG_BLOCK * it];
    }
#ifdef GEMM_PROBE
    NG_STAMP(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NG_STAMP(6);
#endif
#pragma unroll
    for (int it = 0; it < TRIPS; ++it) {
        const int i = i0 + (NG_BLOCK / 32) * it;
        if (k4 < 32 * (i / 32 + 1)) {
            float *d = Ls + i * NG_LDA + k4;
            d[0] = (k4 + 0 <= i) ? v[it].x : 0.0f;
            d[1] = (k4 + 1 <= i) ? v[it].y : 0.0f;
            d[2] = (k4 + 2 <= i) ? v[it].z : 0.0f;
            d[3] = (k4 + 3 <= i) ? v[it].w : 0.0f;
        }
    }
}
