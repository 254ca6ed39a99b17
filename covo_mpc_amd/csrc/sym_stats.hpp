// sym_stats.hpp -- what the Sigma chain needs to know about its input before the first product: per-row |.|-sums and the
// diagonal (Gershgorin bounds of A and of A + delta I), |A|_F^2, trace.  Formed per lower 16x16 tile by whoever holds the tile --
// the Hessian's last launch (KD, fused step: the chain then starts one launch later... earlier: no prep launch) or
// ns_prep_kernel (any other caller) -- through THIS function, so both paths leave bit-identical numbers.
#pragma once
#include "wave_reduce.hpp"

struct SymStatsOut {
    double *rpart;  // [128][8]  sum over the 16 columns of column block cb of |A[r][c]|
    double *fpart;  // [36]      sum of squares of a lower tile (off-diagonal tiles counted twice)
    double *diag;   // [128]
    size_t stride;  // doubles between the blocks of consecutive matrices of a batch (all pointers)
    double *flags;  // != null: 64 doubles the producer clears (the barrier flag words of the Sigma chain's persistent launches)
    double *ready;  // != null: one double the producer clears (the chain's "result is there" flag, polled by the concurrent NS launch)
};

// 256 threads (4 waves) hold the lower tile (I >= J) of the symmetric matrix: thread (lane = 16 hi + lo, wave wv) holds
// v = A[16 I + hi + 4 wv][16 J + lo].  EVERY thread of the workgroup must call this (one barrier inside); `active` = holds a value.
__device__ __forceinline__ void sym_tile_stats(bool active, double v, int I, int J, int tile, int lane, int wv, const SymStatsOut &o,
                                               double (*tmp)[17], double *part)
{
    const int lo = lane & 15, hi = lane >> 4, ri = hi + 4 * wv;
    if (active) {
        const double av = fabs(v);
        const double rs = wr::row16_allsum(av);
        if (lo == 0) o.rpart[(16 * I + ri) * 8 + J] = rs;
        tmp[ri][lo] = av;
        const double e = (I != J) ? 2.0 * v * v : v * v;
        const double ws = wr::wave64_allsum(e);
        if (lane == 0) part[wv] = ws;
        if (I == J && ri == lo) o.diag[16 * I + ri] = v;
    }
    __syncthreads();
    if (active) {
        const int t = 64 * wv + lane;
        if (t < 16 && I != J) {  // the mirrored half: rows of block J, columns of block I
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += tmp[r][t];
            o.rpart[(16 * J + t) * 8 + I] = s;
        }
        if (t == 0) o.fpart[tile] = (part[0] + part[1]) + (part[2] + part[3]);
    }
}
