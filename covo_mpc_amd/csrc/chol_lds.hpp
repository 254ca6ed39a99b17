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

// lower-triangle tile enumeration t -> (a, b), a >= b, a(a+1)/2 + b = t  (t < 28)
__device__ __forceinline__ void chol_tri(int t, int &a, int &b)
{
    a = (t >= 21) ? 6 : (t >= 15) ? 5 : (t >= 10) ? 4 : (t >= 6) ? 3 : (t >= 3) ? 2 : (t >= 1) ? 1 : 0;
    b = t - a * (a + 1) / 2;
}

template <int ld>  // compile-time stride: every LDS address is base + immediate
__device__ void chol128_lds_mfma(double *A, int tid)
{
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
        long long c1 = 0, c2 = 0;
        if (wave < 4) {
            // ---- panel factorisation: waves 0..3 (one per SIMD: the f64 MFMA pipe is the shared resource),
            // each with its own copy of the diagonal block and up to two tiles below it
            const int tiA = p + 1 + wave, tiB = p + 5 + wave;
            const bool hasA = tiA < 8, hasB = tiB < 8;
            chol_f64x4 D, Ta = {0.0, 0.0, 0.0, 0.0}, Tb = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = hi + 4 * r;
                D[r] = A[(j0 + min(row, lo)) * ld + j0 + max(row, lo)];   // symmetric fill from the lower triangle
                if (hasA) Ta[r] = A[(j0 + row) * ld + 16 * tiA + lo];     // tile^T: (k = row, i = lo)
                if (hasB) Tb[r] = A[(j0 + row) * ld + 16 * tiB + lo];
            }
            CHOL_USE(D[0]); CHOL_USE(Ta[0]); CHOL_USE(Tb[0]);
            c1 = CHOL_CLOCK();
            chol_f64x4 LD_ = {0.0, 0.0, 0.0, 0.0}, LA = LD_, LB = LD_;  // the finished columns, in tile layout
            double a_prev = 0.0, ta_prev = 0.0, tb_prev = 0.0, inv_prev = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int hj = j & 3, rj = j >> 2;
                const double piv = wr::bcast_lane(D[rj], 16 * hj + j);
                if (j > 0) {  // the tiles run one column behind: their MFMAs fill the pipe while the rsqrt chain runs
                    Ta = __builtin_amdgcn_mfma_f64_16x16x4f64(a_prev, ta_prev, Ta, 0, 0, 0);
                    Tb = __builtin_amdgcn_mfma_f64_16x16x4f64(a_prev, tb_prev, Tb, 0, 0, 0);
                }
                // 1/sqrt(piv) restricted to the lanes of group hj (0 elsewhere): v_rsq_f64 + one Newton step
                const double y = __builtin_amdgcn_rsq(piv);
                const double ym = y * maskg[hj], yhm = 0.5 * ym;
                const double e = fma(-(piv * y), y, 1.0);
                const double inv_m = fma(yhm, e, ym);
                const double lrow = D[rj] * inv_m;                            // L[j0+lo][j0+j] in group hj, 0 elsewhere
                const double a_op = D[rj] * (inv_m * ((lo > j) ? -1.0 : 0.0));  // rows > j only
                LD_[rj] += lrow;
                D = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op, lrow, D, 0, 0, 0);
                if (j > 0) {  // column j-1 of the tiles has its final value now
                    (void)inv_prev;
                }
                const double ta = Ta[rj] * inv_m, tb = Tb[rj] * inv_m;        // L[16ti+lo][j0+j]
                LA[rj] += ta;
                LB[rj] += tb;
                a_prev = a_op;
                ta_prev = ta;
                tb_prev = tb;
                inv_prev = inv_m;
            }
            CHOL_USE(D[0]); CHOL_USE(Ta[0]); CHOL_USE(Tb[0]);
            c2 = CHOL_CLOCK();
            // rows j = hi + 4r of the finished columns sit in group hi: plain tile stores (the diagonal block's
            // upper triangle receives don't-care values)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = hi + 4 * r;
                if (wave == 0) A[(j0 + row) * ld + j0 + lo] = LD_[r];
                if (hasA) A[(j0 + row) * ld + 16 * tiA + lo] = LA[r];
                if (hasB) A[(j0 + row) * ld + 16 * tiB + lo] = LB[r];
            }
        }
        __syncthreads();
        const long long c3 = CHOL_CLOCK();
        // ---- trailing tiles (ui >= uj > p), round-robin over the 8 waves, two tiles in flight per wave.  The MFMA
        // computes the TRANSPOSED tile D[m][n] = sum_k L[16uj+m][k] L[16ui+n][k] so that lanes (n = lo) run
        // down a column of A in LDS.
        const int T = 7 - p, ntiles = T * (T + 1) / 2;
        for (int t = wave; t < ntiles; t += 16) {
            const bool two = t + 8 < ntiles;
            int a0, b0, a1, b1;
            chol_tri(t, a0, b0);
            chol_tri(two ? t + 8 : t, a1, b1);
            const int ui0 = p + 1 + a0, uj0 = p + 1 + b0, ui1 = p + 1 + a1, uj1 = p + 1 + b1;
            chol_f64x4 acc0, acc1;
            double av0[4], bv0[4], av1[4], bv1[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                av0[kk] = -A[(j0 + 4 * kk + hi) * ld + 16 * uj0 + lo];  // A-operand [m = lo][k = hi]
                bv0[kk] = A[(j0 + 4 * kk + hi) * ld + 16 * ui0 + lo];   // B-operand [k = hi][n = lo]
                av1[kk] = -A[(j0 + 4 * kk + hi) * ld + 16 * uj1 + lo];
                bv1[kk] = A[(j0 + 4 * kk + hi) * ld + 16 * ui1 + lo];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0[r] = A[(16 * uj0 + hi + 4 * r) * ld + 16 * ui0 + lo];
                acc1[r] = A[(16 * uj1 + hi + 4 * r) * ld + 16 * ui1 + lo];
            }
            if (two) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av0[kk], bv0[kk], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av1[kk], bv1[kk], acc1, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av0[kk], bv0[kk], acc0, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                A[(16 * uj0 + hi + 4 * r) * ld + 16 * ui0 + lo] = acc0[r];
                if (two) A[(16 * uj1 + hi + 4 * r) * ld + 16 * ui1 + lo] = acc1[r];
            }
        }
        __syncthreads();
        CHOL_STAMP(0, c2 - c1);
        CHOL_STAMP(16 + p, c2 - c1);
        { const long long c4 = CHOL_CLOCK(); (void)c4; CHOL_STAMP(1, c4 - c3); CHOL_STAMP(8 + p, c4 - c3); }
        CHOL_STAMP(2, c1 - c0);
        CHOL_STAMP(3, c3 - c2);
    }
}
