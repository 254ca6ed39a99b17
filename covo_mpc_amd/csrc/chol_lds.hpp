// chol_lds.hpp -- single-workgroup Cholesky of a small SPD matrix held in LDS (gfx950, fp64).
#pragma once
#include <hip/hip_runtime.h>
#include "wave_reduce.hpp"

// in-LDS lower Cholesky of an n x n SPD matrix (n multiple of 8, n <= 128), column-major with stride ld.
// Right-looking by panels of 8 columns: (1) wave 0 factors the panel in registers (lane l owns rows l and
// l+64; pivots travel by v_readlane), (2) every wave applies the rank-8 update of the panel to its share of
// the trailing columns (8 independent FMAs per entry -- short dependency chains; a left-looking variant
// with one long LDS dot product per entry was latency-bound at 70 us).  Two barriers per panel.
// 1/sqrt(pivot) is seeded with v_rsq_f32 and polished in fp64.
__device__ __forceinline__ double rsqrt_f64(double x)
{
    double y = (double)__builtin_amdgcn_rsqf((float)x);
    y = y * fma(-0.5 * x, y * y, 1.5);
    y = y * fma(-0.5 * x, y * y, 1.5);
    return y;
}
__device__ void chol_lds_fast(double *A, int n, int ld, int tid, int nthreads)
{
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += 8) {
        // (1) factor the panel: wave 0, rows in registers
        if (tid < 64) {
            double P[2][8];
            const int sj = j0 >> 6;  // slot (0: rows 0..63, 1: rows 64..127) holding the panel's diagonal rows
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                P[0][c] = (lane < n) ? A[(j0 + c) * ld + lane] : 0.0;
                P[1][c] = (lane + 64 < n) ? A[(j0 + c) * ld + lane + 64] : 0.0;
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = j0 + jj, lj = j & 63;
                const double diag = wr::bcast_lane(sj ? P[1][jj] : P[0][jj], lj);
                const double inv = rsqrt_f64(diag);
                const double d = diag * inv;  // sqrt(diag)
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const int row = lane + 64 * sl;
                    P[sl][jj] = (row == j) ? d : P[sl][jj] * inv;
                }
#pragma unroll
                for (int c = jj + 1; c < 8; ++c) {
                    const double lc = wr::bcast_lane(sj ? P[1][jj] : P[0][jj], (j0 + c) & 63);  // L[j0+c][j]
                    P[0][c] = fma(-P[0][jj], lc, P[0][c]);
                    P[1][c] = fma(-P[1][jj], lc, P[1][c]);
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (lane >= j0 + c && lane < n) A[(j0 + c) * ld + lane] = P[0][c];
                if (lane + 64 >= j0 + c && lane + 64 < n) A[(j0 + c) * ld + lane + 64] = P[1][c];
            }
        }
        __syncthreads();
        // (2) trailing update: A[c][i] -= sum_{k in panel} L[i][k] L[c][k]   for c >= j0+8, i >= c
        for (int c = j0 + 8 + wave; c < n; c += nwaves) {
            double lc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) lc[k] = A[(j0 + k) * ld + c];
            for (int i = c + lane; i < n; i += 64) {
                double acc = A[c * ld + i];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc = fma(-A[(j0 + k) * ld + i], lc[k], acc);
                A[c * ld + i] = acc;
            }
        }
        __syncthreads();
    }
}

// ---- 128 x 128, 512 threads: right-looking by panels of 16 columns, EVERYTHING on the matrix cores.
// The serial part of a Cholesky is one rsqrt chain per column; the version above spends ~700 cycles per
// column around it (a lone wave, 30 v_readlane + 30 fp64 FMA per column).  Here a 16x16 block lives in the
// MFMA C/D layout (lane = 16*hi + lo holds rows hi+4r, column lo), where
//   * the diagonal block is kept fully symmetric, so "column j" is also ROW j = register j>>2 of the 16
//     lanes with hi = j&3, indexed by lo -- exactly the shape of an MFMA operand with k = hi;
//   * the rank-1 update C -= l l^T is ONE v_mfma_f64_16x16x4_f64 whose A and B operands are that register
//     (times 1/sqrt(pivot), zero outside hi = j&3, A masked to lo > j): no cross-lane traffic at all;
//   * the tiles below the diagonal block are held TRANSPOSED, so their column j is again a register row
//     and the triangular solve is the same rank-1 MFMA with the diagonal block's vector as A operand.
// Every wave factors the diagonal block redundantly (15 instructions per column) next to its own tile
// below it; per column: v_readlane(pivot) -> rsqrt -> 2 multiplies -> 2 MFMA  (~180 cycles, 9 us for the
// 128 columns), then the 8 waves apply the rank-16 update to the trailing tiles (4 MFMA per tile).
// Column-major with stride ld (element (r, c) at A[c*ld + r]); only the lower triangle is referenced/valid.
typedef double chol_f64x4 __attribute__((ext_vector_type(4)));
#ifdef CHOL_PROBE
__device__ long long chol_prof[32];
#define CHOL_STAMP(i, expr) do { if (tid == 0) chol_prof[i] += (expr); } while (0)
__device__ __forceinline__ long long chol_tick() { __builtin_amdgcn_sched_barrier(0); long long t = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); return t; }
#define CHOL_CLOCK() chol_tick()
#define CHOL_USE(v) asm volatile("" ::"v"(v))
#else
#define CHOL_STAMP(i, expr) do { (void)c0; (void)c1; (void)c2; (void)c3; } while (0)
#define CHOL_CLOCK() 0
#define CHOL_USE(v) do { } while (0)
#endif

// 1/sqrt(x): v_rsq_f64 (measured 5e-8 relative on gfx950) + one Newton step -> 4e-15; on the per-column
// critical path of the factorisation (5 instructions instead of 10 for the f32-seeded version above)
__device__ __forceinline__ double rsqrt_f64_nr1(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    return fma(0.5 * y, e, y);
}

// Left-looking by 16-column panels, three phases per panel, two barriers:
//   U  wave w takes tile (p + w, p) of the panel (w = 0: the diagonal block) and applies ALL previous panels to it:
//      tile -= L(i, q) L(p, q)^T for q < p, 4 p MFMAs on an accumulator that stays in registers (a right-looking
//      version re-reads and re-writes every trailing tile once per panel through LDS: 19k of its 64k cycles);
//   F  wave 0 factors the diagonal block column by column (rank-1 MFMA updates, see above) and carries the SAME
//      operations on an identity tile, which ends up as W = L_pp^-1 -- two MFMAs per column on one SIMD's pipe
//      (factoring the block redundantly in every wave next to its own tile made the pipe the bottleneck: 285
//      cycles per column);
//   T  the other waves finish their tiles with one 16x16x16 product  L(i, p)^T = W . tile^T  (4 MFMAs).
// Column-major with stride ld (element (r, c) at A[c*ld + r]); the input must be the FULL symmetric matrix (the
// diagonal blocks are read as stored); on return the lower triangle holds L.
// tile^T (panel pc, tile row ti) -= L(pc, q) L(ti, q)^T  as four MFMAs; Tt in C/D layout (register g = panel column 4g + hi)
template <int ld>
__device__ __forceinline__ void chol_tile_update(const double *A, chol_f64x4 &Tt, int q, int pc, int ti, int lo, int hi)
{
    double av[4], bv[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        av[kk] = -A[(16 * q + 4 * kk + hi) * ld + 16 * pc + lo];  // A-operand [m = k = lo][c = hi]: L(pc, q)
        bv[kk] = A[(16 * q + 4 * kk + hi) * ld + 16 * ti + lo];   // B-operand [c = hi][n = r = lo]: L(ti, q)
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) Tt = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], Tt, 0, 0, 0);
}

template <int ld>  // compile-time stride: every LDS address is base + immediate
__device__ void chol128_lds_mfma(double *A, int tid)
{
    __shared__ double Wsm[16 * 16];
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lo = lane & 15, hi = lane >> 4;
    double maskg[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) maskg[g] = (hi == g) ? 1.0 : 0.0;
    __syncthreads();
#ifdef CHOL_PROBE
    if (tid == 0) for (int q = 0; q < 32; ++q) chol_prof[q] = 0;
#endif
    for (int p = 0; p < 8; ++p) {
        const long long c0 = CHOL_CLOCK();
        const int j0 = 16 * p;
        const int ti = p + wave;  // this wave's tile row in the panel
        const bool has = ti < 8;
        // ---- U: tile^T (k = panel column, r = row in tile) in C/D layout: register g, lane (lo = r, hi) = column 4g + hi
        chol_f64x4 Tt = {0.0, 0.0, 0.0, 0.0};
        if (has) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Tt[r] = A[(j0 + hi + 4 * r) * ld + 16 * ti + lo];
            if (p > 0) chol_tile_update<ld>(A, Tt, p - 1, p, ti, lo, hi);  // panels q < p-1 were applied during F(p-1)
        }
        CHOL_USE(Tt[0]);
        const long long c1 = CHOL_CLOCK();
        if (wave == 0) {
            // ---- F: D = Tt (the symmetric diagonal block), W = identity; per column j: pivot -> rsqrt -> scaled row
            chol_f64x4 D = Tt, W, LD_ = {0.0, 0.0, 0.0, 0.0}, LW = LD_;
#pragma unroll
            for (int r = 0; r < 4; ++r) W[r] = (hi + 4 * r == lo) ? 1.0 : 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int hj = j & 3, rj = j >> 2;
                const double piv = wr::bcast_lane(D[rj], 16 * hj + j);
                const double y = __builtin_amdgcn_rsq(piv);
                const double ym = y * maskg[hj], yhm = 0.5 * ym;
                const double e = fma(-(piv * y), y, 1.0);
                const double inv_m = fma(yhm, e, ym);                          // 1/sqrt(piv) in group hj, 0 elsewhere
                const double lrow = D[rj] * inv_m;                             // L[j0+lo][j0+j] in group hj
                const double a_op = D[rj] * (inv_m * ((lo > j) ? -1.0 : 0.0));  // rows > j only
                const double wrow = W[rj] * inv_m;                             // row j of L_pp^-1 (final)
                LD_[rj] += lrow;
                LW[rj] += wrow;
                // both rank-1 updates go out back to back: the wait for the D result then also covers W
                __builtin_amdgcn_sched_barrier(0);
                D = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op, lrow, D, 0, 0, 0);
                W = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op, wrow, W, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                A[(j0 + hi + 4 * r) * ld + j0 + lo] = LD_[r];   // row j = hi + 4r of L^T; the upper triangle gets don't-cares
                Wsm[(hi + 4 * r) * 16 + lo] = LW[r];            // W[k][c]
            }
        } else if (p + wave < 8 && p > 0) {
            // ---- look-ahead (the waves that wait for W): apply the panels q < p, already final, to tile (p + wave, p + 1)
            // of the NEXT panel in place, so that its U phase only has panel p left (4 MFMAs instead of 4 (p + 1))
            const int tn = p + wave;
            chol_f64x4 Nt;
#pragma unroll
            for (int r = 0; r < 4; ++r) Nt[r] = A[(j0 + 16 + hi + 4 * r) * ld + 16 * tn + lo];
            for (int q = 0; q < p; ++q) chol_tile_update<ld>(A, Nt, q, p + 1, tn, lo, hi);
#pragma unroll
            for (int r = 0; r < 4; ++r) A[(j0 + 16 + hi + 4 * r) * ld + 16 * tn + lo] = Nt[r];
        }
        __syncthreads();
        const long long c2 = CHOL_CLOCK();
        // ---- T: L(ti, p)^T = W . tile^T
        if (has && wave > 0) {
            chol_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            double wv[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wv[kk] = Wsm[lo * 16 + 4 * kk + hi];  // A-operand [m = k = lo][c = hi]
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[kk], Tt[kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) A[(j0 + hi + 4 * r) * ld + 16 * ti + lo] = acc[r];
        }
        __syncthreads();
        const long long c3 = CHOL_CLOCK();
        CHOL_STAMP(0, c2 - c1);
        CHOL_STAMP(1, c3 - c2);
        CHOL_STAMP(2, c1 - c0);
        CHOL_STAMP(3, 0);
        CHOL_STAMP(8 + p, c1 - c0);
        CHOL_STAMP(16 + p, c2 - c1);
    }
}
