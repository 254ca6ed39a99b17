// softmax_merge.hpp -- stage 2 of the exp(-cost/lambda)-weighted aggregation (covo.py:266-275) as a device function: merges G
// online-softmax records {m, s, v[128]} into the new mean (or into one merged record).  ONE body for its three callers, so that
// they agree bit for bit whatever their workgroup size:
//   merge_kernel (reduce.hip, 1 024 threads)          the stand-alone launch; also merges the all-gathered rank records
//   the last workgroup of the fused small step         (step_small.hip, 256 threads: the launch finishes its own update)
//   the last workgroup of the record-leaving rollout   (rollout_pipe.hpp, 192 .. 768 threads, fused single-GPU steps)
// The arithmetic is laid out over VIRTUAL lanes, not over the caller's threads: record g is "thread g" of a 1 024-thread
// workgroup (its 64-record virtual waves are summed with the wave butterfly, the 16 wave sums in ascending order), column c of
// slice q sums the records g = q, q + 8, q + 16, ... in ascending order with one fma each, the 8 slices are added in ascending
// order.  A caller with fewer threads walks the same virtual lanes in several rounds.  Fixed order: bit-reproducible, and
// independent of which workgroup happens to run it.
// COH: the records were written by OTHER workgroups of the SAME launch (agent-scope relaxed atomic stores, write-through): read
// them with agent-scope relaxed atomic loads (sc1: served coherently across the XCDs' L2s, no cache-wide fence).
#pragma once
#include "covo_common.hpp"

constexpr int MG_THREADS = 1024;
constexpr int MG_SLICES = MG_THREADS / COVO_NA;  // 8
constexpr int MG_MAXG = 1024;
constexpr int MG_VWAVES = MG_THREADS / 64;       // 16

struct MergeLds {
    float scale[MG_MAXG];
    float redm[MG_VWAVES];
    float reds[MG_VWAVES];
    float sv[MG_SLICES][COVO_NA];
};

template <bool COH>
__device__ __forceinline__ float mg_load(const float *p)
{
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// FINAL: out[128] = gamma * v / s + (1 - gamma) * a_mean_old (covo.py:270-275); else out[130] = the merged record {m, s, v}.
// stride: floats between consecutive records.  Every thread of the workgroup calls it (it contains barriers); THREADS is a
// multiple of 64.  G <= MG_MAXG.
template <int THREADS, bool FINAL, bool COH>
__device__ __forceinline__ void merge_body(const float *__restrict__ partials, int G, float inv_lam, const float *__restrict__ a_mean_old,
                                           float gamma_mean, float *__restrict__ out, int stride, MergeLds &L)
{
#pragma clang fp contract(off)  // every fused multiply-add below is written out: the same bits in every translation unit
    static_assert(THREADS % 64 == 0 && THREADS <= MG_THREADS, "merge_body: THREADS");
    constexpr int ROUNDS = (MG_THREADS + THREADS - 1) / THREADS;  // virtual threads per thread
    constexpr int NW = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // phase 3's operands of this thread's first virtual lane first (they depend on nothing: their round trip overlaps phases 1
    // and 2 -- the records other XCDs have just written are most of this merge's time)
    constexpr int MG_PRE = 32;
    float vals[MG_PRE];
    {
        const int col = tid & (COVO_NA - 1), slice = tid >> 7;  // (tid < 1 024: slice < 8)
#pragma unroll
        for (int i = 0; i < MG_PRE; ++i) {
            const int g = slice + MG_SLICES * i;
            vals[i] = (g < G) ? mg_load<COH>(partials + (size_t)g * stride + 2 + col) : 0.0f;
        }
    }
    // phase 1: m = min_g m_g (exact in any order); a virtual thread's header {m_g, s_g} stays in registers for phase 2
    float my_m[ROUNDS], my_s[ROUNDS];
    float m = __builtin_inff();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int g = tid + r * THREADS;
        const bool mine = g < G && g < MG_THREADS;
        my_m[r] = mine ? mg_load<COH>(partials + (size_t)g * stride) : __builtin_inff();
        my_s[r] = mine ? mg_load<COH>(partials + (size_t)g * stride + 1) : 0.0f;
        m = fminf(m, my_m[r]);
    }
    m = wave_min(m);
    if (lane == 0) L.redm[wave] = m;
    if (tid < MG_VWAVES) L.reds[tid] = 0.0f;
    __syncthreads();
    m = L.redm[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) m = fminf(m, L.redm[i]);
    // phase 2: per-record scale; s = sum_g s_g scale_g as 16 virtual-wave butterflies, then in ascending wave order
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int g = tid + r * THREADS;  // THREADS % 64 == 0: a wave holds one whole virtual wave, lane = g & 63
        if (g - lane < MG_THREADS) {      // (wave-uniform)
            float sg = 0.0f;
            if (g < G) {
                const float sc = (my_s[r] > 0.0f) ? expf((m - my_m[r]) * inv_lam) : 0.0f;  // empty shard -> 0
                L.scale[g] = sc;
                sg = my_s[r] * sc;
            }
            sg = wave_sum(sg);
            if (lane == 0) L.reds[g >> 6] = sg;
        }
    }
    __syncthreads();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < MG_VWAVES; ++i) s += L.reds[i];
    // phase 3: v[col] = sum_g v_g[col] scale_g per slice (ascending g within a slice)
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int vt = tid + r * THREADS;
        if (vt < MG_THREADS) {
            const int col = vt & (COVO_NA - 1), slice = vt >> 7;
            float v = 0.0f;
            if (r == 0) {
#pragma unroll
                for (int i = 0; i < MG_PRE; ++i) {
                    const int g = slice + MG_SLICES * i;
                    if (g < G) v = __builtin_fmaf(vals[i], L.scale[g], v);
                }
#pragma unroll 4
                for (int g = slice + MG_SLICES * MG_PRE; g < G; g += MG_SLICES)
                    v = __builtin_fmaf(mg_load<COH>(partials + (size_t)g * stride + 2 + col), L.scale[g], v);
            } else {
#pragma unroll 8
                for (int g = slice; g < G; g += MG_SLICES)
                    v = __builtin_fmaf(mg_load<COH>(partials + (size_t)g * stride + 2 + col), L.scale[g], v);
            }
            L.sv[slice][col] = v;
        }
    }
    __syncthreads();
    if (tid < COVO_NA) {
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < MG_SLICES; ++i) v += L.sv[i][tid];
        if (FINAL) {
            out[tid] = __builtin_fmaf(gamma_mean, v / s, a_mean_old[tid] * (1.0f - gamma_mean));
        } else {
            out[2 + tid] = v;
            if (tid == 0) {
                out[0] = m;
                out[1] = s;
            }
        }
    }
}
