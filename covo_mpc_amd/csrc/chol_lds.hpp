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
//   F  wave 0 factors the diagonal block four columns at a time (a closed-form 4x4 factor on every lane + rank-4 MFMA
//      updates, see there) and carries the SAME operations on an identity tile, which ends up as W = L_pp^-1
//      (rounds 2-4: column by column with rank-1 MFMA updates, 362 cycles per column; factoring the block
//      redundantly in every wave next to its own tile made the pipe the bottleneck: 285 cycles per column);
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

// NWAVES: 8 (one wave per tile row of a panel: the finalize launch's 512-thread workgroup) or 4 (a 256-thread workgroup: wave 0
// factors, waves 1-3 take up to three tiles each -- the log-det factorisation riding in the Newton-Schulz launch, sigma_ns.hip);
// the same tile arithmetic either way.
struct CholNoHook {
    __device__ __forceinline__ void panel_done(int) const {}
    __device__ __forceinline__ void before_barrier(int) const {}
};
// hook.panel_done(p): called by every thread once panel p (columns 16 p .. 16 p + 15 of L, all rows) is final in LDS and will not
// be written again -- the finalize launch's streamed variant sends it on from there while the next panel is factored;
// hook.before_barrier(p): called by every thread of panel p's iteration right before the barrier that ends its F / look-ahead
// phase (the place to wait for what panel_done(p - 1) started without holding anybody up).
template <int ld, int NWAVES = 8, class Hook = CholNoHook>  // compile-time stride: every LDS address is base + immediate
__device__ void chol128_lds_mfma(double *A, int tid, Hook hook = Hook())
{
    constexpr int NW1 = NWAVES - 1, U = (7 + NW1 - 1) / NW1;  // worker waves, tiles per worker wave and panel
    __shared__ double Wsm[16 * 16];
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lo = lane & 15, hi = lane >> 4;
    __syncthreads();
#ifdef CHOL_PROBE
    if (tid == 0) for (int q = 0; q < 32; ++q) chol_prof[q] = 0;
#endif
    for (int p = 0; p < 8; ++p) {
        const long long c0 = CHOL_CLOCK();
        const int j0 = 16 * p;
        // ---- U: tile^T (k = panel column, r = row in tile) in C/D layout: register g, lane (lo = r, hi) = column 4g + hi
        // wave 0: the diagonal tile; worker wave w, slot u: tile row p + 1 + (w - 1) + NW1 u
        chol_f64x4 TtU[U];
        bool hasU[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ti = (wave == 0) ? p : p + wave + NW1 * u;
            hasU[u] = ti < 8 && (wave != 0 || u == 0);
            TtU[u] = chol_f64x4{0.0, 0.0, 0.0, 0.0};
            if (hasU[u]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) TtU[u][r] = A[(j0 + hi + 4 * r) * ld + 16 * ti + lo];
                if (p > 0) chol_tile_update<ld>(A, TtU[u], p - 1, p, ti, lo, hi);  // panels q < p-1 were applied during F(p-1)
            }
        }
        chol_f64x4 &Tt = TtU[0];
        CHOL_USE(Tt[0]);
        const long long c1 = CHOL_CLOCK();
        if (wave == 0) {
            // ---- F: D = Tt (the symmetric diagonal block), W = identity, four columns at a time (round 5).
            // Rounds 2-4 factored column by column: pivot -> rsqrt -> scaled row -> two rank-1 MFMAs (D and W), 362 cycles per
            // column, of which the two 64-cycle fp64 MFMAs and ~20 dependent fp64 VALU instructions add up on the one SIMD.  A
            // 16x16x4 MFMA is a rank-FOUR update, and the layout already keeps column j in the lanes with hi = j & 3: the four
            // columns 4 mm .. 4 mm + 3 ARE register mm.  So per micro-panel mm:
            //   1. the 4x4 diagonal micro-block T (10 distinct entries of register mm) travels to all lanes by v_readlane;
            //   2. every lane factors it, T = Lt Lt^T, and inverts the factor, M = Lt^-1 (closed form: 4 rsqrt chains);
            //   3. ONE MFMA forms the micro-panel's final columns for all 16 rows: Lpan^T = M . Spanel^T -- A operand M (lanes
            //      lo < 4), B operand register mm as it stands; the result row m = hi lands in C/D register 0 at lane (lo, hi)
            //      = l[lo][4 mm + hi]: exactly the operand layout of the rank-4 update.  The same product on W's register mm
            //      gives the four final rows of L_pp^-1;
            //   4. ONE MFMA each applies the rank-4 update to D and to W (A operand masked to the rows below each column).
            // 4 MFMAs per 4 columns instead of 8, and the serial chain per column is a quarter of {readlane, 4x4 factor, 2 MFMAs}.
            chol_f64x4 D = Tt, W, LD_ = {0.0, 0.0, 0.0, 0.0}, LW = LD_;
            const chol_f64x4 Zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) W[r] = (hi + 4 * r == lo) ? 1.0 : 0.0;
            const int mcode = (lo < 4 && hi <= lo) ? 4 * lo + hi : -1;  // which entry of M this lane feeds into the A operand
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                // T[a][b] = S[4 mm + a][4 mm + b] sits in register mm at lane (lo = 4 mm + a, hi = b)
                const double t00 = wr::bcast_lane(D[mm], 16 * 0 + 4 * mm + 0);
                const double t10 = wr::bcast_lane(D[mm], 16 * 0 + 4 * mm + 1);
                const double t20 = wr::bcast_lane(D[mm], 16 * 0 + 4 * mm + 2);
                const double t30 = wr::bcast_lane(D[mm], 16 * 0 + 4 * mm + 3);
                const double t11 = wr::bcast_lane(D[mm], 16 * 1 + 4 * mm + 1);
                const double t21 = wr::bcast_lane(D[mm], 16 * 1 + 4 * mm + 2);
                const double t31 = wr::bcast_lane(D[mm], 16 * 1 + 4 * mm + 3);
                const double t22 = wr::bcast_lane(D[mm], 16 * 2 + 4 * mm + 2);
                const double t32 = wr::bcast_lane(D[mm], 16 * 2 + 4 * mm + 3);
                const double t33 = wr::bcast_lane(D[mm], 16 * 3 + 4 * mm + 3);
                // Lt (below the diagonal) and the reciprocals of its diagonal
                const double i0 = rsqrt_f64_nr1(t00);
                const double l10 = t10 * i0, l20 = t20 * i0, l30 = t30 * i0;
                const double i1 = rsqrt_f64_nr1(fma(-l10, l10, t11));
                const double l21 = fma(-l20, l10, t21) * i1, l31 = fma(-l30, l10, t31) * i1;
                const double i2 = rsqrt_f64_nr1(fma(-l21, l21, fma(-l20, l20, t22)));
                const double l32 = fma(-l31, l21, fma(-l30, l20, t32)) * i2;
                const double i3 = rsqrt_f64_nr1(fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, t33))));
                // M = Lt^-1 (lower)
                const double m10 = -(l10 * i0) * i1;
                const double m21 = -(l21 * i1) * i2;
                const double m20 = -fma(l21, m10, l20 * i0) * i2;
                const double m32 = -(l32 * i2) * i3;
                const double m31 = -fma(l32, m21, l31 * i1) * i3;
                const double m30 = -fma(l32, m20, fma(l31, m10, l30 * i0)) * i3;
                double aop = 0.0;
                aop = (mcode == 0) ? i0 : aop;
                aop = (mcode == 4) ? m10 : aop;
                aop = (mcode == 5) ? i1 : aop;
                aop = (mcode == 8) ? m20 : aop;
                aop = (mcode == 9) ? m21 : aop;
                aop = (mcode == 10) ? i2 : aop;
                aop = (mcode == 12) ? m30 : aop;
                aop = (mcode == 13) ? m31 : aop;
                aop = (mcode == 14) ? m32 : aop;
                aop = (mcode == 15) ? i3 : aop;
                __builtin_amdgcn_sched_barrier(0);
                const chol_f64x4 P = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, D[mm], Zero, 0, 0, 0);  // P[0](lo, hi) = L[j0 + lo][j0 + 4 mm + hi]
                const chol_f64x4 Q = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, W[mm], Zero, 0, 0, 0);  // Q[0](lo, hi) = L_pp^-1[4 mm + hi][lo]
                const double lpan = P[0], wpan = Q[0];
                const double a_op = (lo - hi > 4 * mm) ? -lpan : 0.0;  // column 4 mm + hi acts on the rows below it only
                LD_[mm] = lpan;
                LW[mm] = wpan;
                if (mm < 3) {  // (the last micro-panel leaves nothing to update)
                    D = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op, lpan, D, 0, 0, 0);
                    W = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op, wpan, W, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                A[(j0 + hi + 4 * r) * ld + j0 + lo] = LD_[r];   // row j = hi + 4r of L^T; the upper triangle gets don't-cares
                Wsm[(hi + 4 * r) * 16 + lo] = LW[r];            // W[k][c]
            }
        } else if (p > 0) {
            // ---- look-ahead (the waves that wait for W): apply the panels q < p, already final, to the tiles (tn, p + 1)
            // of the NEXT panel in place, so that its U phase only has panel p left (4 MFMAs instead of 4 (p + 1))
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int tn = p + wave + NW1 * u;
                if (tn < 8) {
                    chol_f64x4 Nt;
#pragma unroll
                    for (int r = 0; r < 4; ++r) Nt[r] = A[(j0 + 16 + hi + 4 * r) * ld + 16 * tn + lo];
                    for (int q = 0; q < p; ++q) chol_tile_update<ld>(A, Nt, q, p + 1, tn, lo, hi);
#pragma unroll
                    for (int r = 0; r < 4; ++r) A[(j0 + 16 + hi + 4 * r) * ld + 16 * tn + lo] = Nt[r];
                }
            }
        }
        hook.before_barrier(p);
        __syncthreads();
        const long long c2 = CHOL_CLOCK();
        // ---- T: L(ti, p)^T = W . tile^T
        if (wave > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (hasU[u]) {
                    const int ti = p + wave + NW1 * u;
                    chol_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
                    double wv[4];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) wv[kk] = Wsm[lo * 16 + 4 * kk + hi];  // A-operand [m = k = lo][c = hi]
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[kk], TtU[u][kk], acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) A[(j0 + hi + 4 * r) * ld + 16 * ti + lo] = acc[r];
                }
            }
        }
        __syncthreads();
        const long long c3 = CHOL_CLOCK();
        CHOL_STAMP(0, c2 - c1);
        CHOL_STAMP(1, c3 - c2);
        CHOL_STAMP(2, c1 - c0);
        CHOL_STAMP(3, 0);
        CHOL_STAMP(8 + p, c1 - c0);
        CHOL_STAMP(16 + p, c2 - c1);
        hook.panel_done(p);
    }
}
