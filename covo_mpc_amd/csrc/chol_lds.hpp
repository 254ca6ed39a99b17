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

// ---- 128 x 128, 512 threads: right-looking by panels of 16 columns with the trailing update on the matrix
// cores.  Per panel: (1) wave 0 factors the 16 panel columns for ALL rows below the diagonal in registers
// (lane l owns rows l and l+64; pivot rows travel by v_readlane -- the only serial part: one rsqrt chain
// per column), (2) the 8 waves apply the rank-16 update tile by tile: C(16x16) -= Lp_i . Lp_j^T as four
// v_mfma_f64_16x16x4_f64 per tile, the operands fetched ONCE per tile from LDS (the VALU version above
// re-reads one LDS operand per FMA and is LDS-bandwidth bound: 65 us; this one: ~15 us).
// Column-major with stride ld (element (r, c) at A[c*ld + r]); only the lower triangle is referenced/valid.
typedef double chol_f64x4 __attribute__((ext_vector_type(4)));

template <int SJ>  // SJ = 1: the panel's diagonal rows live in slot 1 (rows 64..127) and slot 0 is finished
__device__ __forceinline__ void chol_panel16(double *A, int ld, int j0, int lane)
{
    double P[2][16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (SJ == 0) P[0][c] = A[(j0 + c) * ld + lane];
        P[1][c] = A[(j0 + c) * ld + lane + 64];
    }
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int j = j0 + jj, lj = j & 63;
        const double diag = wr::bcast_lane(P[SJ][jj], lj);
        const double inv = rsqrt_f64(diag);
        const double d = diag * inv;  // sqrt(diag)
        if (SJ == 0) P[0][jj] = (lane == j) ? d : P[0][jj] * inv;
        P[1][jj] = (lane + 64 == j) ? d : P[1][jj] * inv;
#pragma unroll
        for (int c = jj + 1; c < 16; ++c) {
            const double lc = wr::bcast_lane(P[SJ][jj], (j0 + c) & 63);  // L[j0+c][j]
            if (SJ == 0) P[0][c] = fma(-P[0][jj], lc, P[0][c]);
            P[1][c] = fma(-P[1][jj], lc, P[1][c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (SJ == 0 && lane >= j0 + c) A[(j0 + c) * ld + lane] = P[0][c];
        if (lane + 64 >= j0 + c) A[(j0 + c) * ld + lane + 64] = P[1][c];
    }
}

__device__ void chol128_lds_mfma(double *A, int ld, int tid)
{
    const int lane = tid & 63, wave = tid >> 6, lo = lane & 15, hi = lane >> 4;
    __syncthreads();
    for (int p = 0; p < 8; ++p) {
        const int j0 = 16 * p;
        if (wave == 0) {
            if (j0 < 64) chol_panel16<0>(A, ld, j0, lane);
            else chol_panel16<1>(A, ld, j0, lane);
        }
        __syncthreads();
        // trailing tiles (ti >= tj > p), round-robin over the 8 waves.  The MFMA computes the TRANSPOSED tile
        // D[m][n] = sum_k L[16tj+m][k] L[16ti+n][k] so that lanes (n = lo) run down a column of A in LDS.
        const int T = 7 - p, ntiles = T * (T + 1) / 2;
        for (int t = wave; t < ntiles; t += 8) {
            int a = 0;
            while ((a + 1) * (a + 2) / 2 <= t) ++a;
            const int ti = p + 1 + a, tj = p + 1 + (t - a * (a + 1) / 2);
            chol_f64x4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = A[(16 * tj + hi + 4 * r) * ld + 16 * ti + lo];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const double av = -A[(j0 + 4 * kk + hi) * ld + 16 * tj + lo];  // A-operand [m = lo][k = hi]
                const double bv = A[(j0 + 4 * kk + hi) * ld + 16 * ti + lo];   // B-operand [k = hi][n = lo]
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) A[(16 * tj + hi + 4 * r) * ld + 16 * ti + lo] = acc[r];
        }
        __syncthreads();
    }
}
