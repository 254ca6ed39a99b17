// eps_tiles.hpp -- epsilon in the noise GEMM's B-operand order, so that it can be drawn AHEAD of the GEMM.
//
// The noise draw a = clip(mu + L eps) needs L, the very last product of the covo-online Sigma chain, but eps itself
// (covo.py:212-221: N(0, I) from the step's act key) depends on nothing but the key.  In the fused step the finalize
// kernel of the chain -- ONE workgroup factoring Z for ~30 us while 255 CUs idle -- carries extra workgroups that
// draw the whole (N, 128) epsilon into HBM/Infinity Cache (parallel branches of a hipGraph were measured to cost more
// than they hide); the GEMM that follows then only loads it.  Layout: wave-tile t = samples
// [32 t, 32 t + 32); for chunk q = 4 g + i (g = k-group, i = 0..3) lane (j = lane & 31, kh = lane >> 5) owns the
// float4 normal4(2 q + kh, sample 32 t + j) -- exactly the register image noise_gemm_kernel builds (BTile), stored as
// eps_tiled[(16 t + q) * 64 + lane]: every load / store instruction of a wave moves one contiguous 1 KiB.
// Same Philox counters as everywhere else (rng_device.hpp): the values are those of covo_randn.
#pragma once
#include "rng_device.hpp"

struct EpsGenArgs {
    float4 *eps_tiled;      // [ceil(N/32)][16][64] float4, null = no generation
    const uint32_t *dyn;    // {key0, key1} in device memory (step.hip: the act key derived by step_begin_kernel)
    int64_t sample_offset;  // global id of local sample 0
    int N;
    // env-batched step: n_inst instances of N samples each; instance e has its key at dyn + e * dyn_stride and its tiles at
    // eps_tiled + e * eps_stride (single step: 1, 0, 0)
    int n_inst, dyn_stride;
    size_t eps_stride;
};

__device__ __forceinline__ void eps_tiles_generate(const EpsGenArgs &G, int wave_global, int wave_stride, int lane)
{
    const int ntiles = (G.N + 31) / 32;
    const int j = lane & 31, kh = lane >> 5;
    for (int tt = wave_global; tt < ntiles * G.n_inst; tt += wave_stride) {
        const int e = tt / ntiles, t = tt - e * ntiles;
        const uint32_t k0 = G.dyn[(size_t)e * G.dyn_stride], k1 = G.dyn[(size_t)e * G.dyn_stride + 1];
        int row = t * 32 + j;
        row = row < G.N ? row : G.N - 1;
        const uint64_t id = (uint64_t)(G.sample_offset + row);
        float4 *out = G.eps_tiled + (size_t)e * G.eps_stride + (size_t)t * 16 * 64 + lane;
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const float4 v = rngd::normal4((uint32_t)(2 * q + kh), id, k0, k1);
            // write-through stores (sc0 sc1): none of the 33 MB stays dirty in the L2s for the end-of-kernel write-back behind
            // the one factoring workgroup (finalize 37 -> 31 us; the GEMM then reads epsilon from HBM instead of L2, so the step
            // gains only ~0.7 us over plain stores: -DEPS_STORE_PLAIN)
#ifndef EPS_STORE_PLAIN
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v vv = {v.x, v.y, v.z, v.w};
            // s_nop 1: a VALU write to the data registers of a > 64-bit store needs two wait states on gfx940+ (the store reads its
            // upper dwords late); hipcc's hazard recognizer inserts them for its own stores but cannot see into inline asm -- without
            // them the next draw's arithmetic clobbered z / w of a few samples whenever the scheduler put it right behind the store
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(out + q * 64), "v"(vv) : "memory");
#else
            out[q * 64] = v;
#endif
        }
    }
}
