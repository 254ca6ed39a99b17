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
