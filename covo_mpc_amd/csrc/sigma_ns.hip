// sigma_ns.hip -- CoVO's optimal covariance WITHOUT an eigendecomposition (gfx950, fp64 MFMA).
//
// Replaces quadjax/controllers/covo.py:116-132 (optimize_sigma) + the Cholesky factorisation inside
// jax.random.multivariate_normal (covo.py:216).  The reference computes, with R symmetrised,
//     eigns, u = eigh(R);  o = eigns - min(eigns) + 1e-2;  Sigma = u diag(exp(log_s)) u^T,
//     log_s = 0.5*log_const - 0.5*log(o),   log_const = (2 n 2 log(sigma) + sum log o)/n,
// i.e. the MATRIX FUNCTION   Sigma = c (R + delta I)^(-1/2),   delta = 1e-2 - lambda_min(R),
//     log c = 2 log(sigma) + log det(R + delta I) / (2 n)        (so that det Sigma = sigma^(2n)).
// A 128x128 eigensolver is a latency monster on a GPU (cyclic Jacobi on one CU: 10-12 sweeps on
// CoVO Hessians, whose spectrum clusters near 0 -> ~2 ms; see sigma.hip, kept as covo_sigma_jacobi).
// The same matrix is obtained here from a chain of ~45 tiny DEPENDENT launches, each one 128^3 fp64 GEMM
// (v_mfma_f64_16x16x4_f64).  What one link of such a chain costs was measured first
// (scripts/probe/chain_probe.hip, profiles/r01_chain_probe.log): an empty graph node 1.6 us; one wave per
// 16x16 tile with K = 128: 4.2 us; one 256-thread workgroup per tile with K split over its 4 waves:
// 2.6 us; the same phases inside ONE persistent launch separated by a counter barrier and fences: 6.1 us (spread
// over the XCDs) / 22.8 us (confined to one XCD) -- so: separate launches, K-split tiles (round 1).  Rounds 2-4 found the
// persistent form that does pay (no fences: sc1 accesses; then one XCD, plain stores, flag words): see ns_flag_barrier.
//   1. lambda_min:  a Chebyshev filter of degree 2^k built by repeated squaring; the Rayleigh-Ritz step reads the FIRST iterate whose
//      bottom pair passes its own residual test (round 5: ritz_eval / ritz_decide, evaluated inside the squaring launch).  Y0 = affine map of A that sends
//      [cut, hi] to [-1, 1] and everything BELOW cut above 1 (hi >= lambda_max: min of the Gershgorin and Frobenius
//      bounds; cut = min_i A_ii + 2^-10 (hi - min_i A_ii) > lambda_min, a Rayleigh quotient).  T_2(x) = 2x^2 - 1
//      composed k times is T_(2^k): eigenvalues inside [cut, hi] stay bounded by 1 while those below grow like
//      cosh(2^k sqrt(2 eps)), eps = their relative distance below cut -- the square root is what a plain power
//      iteration on hi*I - A lacks: closed-loop CoVO Hessians are indefinite with lambda_max ~ 10^2..10^3 and
//      bottom gaps ~ 0.5 (relative 1e-3), where X <- X^2 needed 14-16 squarings (and hit its cap of 16 in 44 % of the
//      steps of an episode) and the filter needs 10-12 (scripts/ns_counts.py).  In normalised form
//          X_(k+1) = X_k^2 / |X_k|_F^2 - I / t_(k+1),   t_(k+1) = 2 t_k^2 |X_k|_F^2,  t_0 = 1  (T_(2^k) = t_k X_k),
//      so once t_k is large the iteration IS the old normalised squaring and X tends to the projector onto the bottom
//      eigenspace; a Rayleigh-Ritz step on its RITZ largest-diagonal columns gives lambda_min
//      (exact as soon as the eigenvector lies in the span -- robust to near-degenerate bottoms).
//   2. B^(-1/2), B = R + delta I: coupled Newton-Schulz  T = a_k I + b_k Z Y, Y <- Y T, Z <- T Z  from
//      Y0 = B/s, Z0 = I, with the Chen-Chow scaling  a_k = 1.5 rho_k, b_k = -0.5 rho_k^3,
//      rho_k^2 = 3/(1 + l_k + l_k^2),  l_{k+1} = l_k (a_k + b_k l_k^2),  l_0 = sqrt(1e-2/s)  -- lambda_min(B)
//      is 1e-2 BY CONSTRUCTION, so the spectrum interval of every iterate is known a priori and the slow
//      initial phase of the unscaled iteration (x1.5 per step) becomes x2.6 per step: 9-10 iterations
//      instead of 17-18 on CoVO Hessians (s/1e-2 ~ 2e4).  Every iterate is stored together with its
//      transpose (both MFMA operands are then read as 128-B runs); symmetry is neither assumed nor forced
//      (mirroring the lower triangle makes the coupled iteration blow up after convergence).
//      Iteration 0 needs no product at all (Z0 = I, and Y0^2 is rebuilt from the filter's first iterate X_1): Y1 = a0 Y0 + b0 Y0^2,
//      Z1 = a0 I + b0 Y0 are written out element-wise (NsFirst, round 5).
//      DEFLATION of the bottom eigenpair (round 3).  By construction B's smallest eigenvalue is 1e-2 -- but only ONE eigenvalue
//      sits there: CoVO Hessians have ~58 negative eigenvalues spread over [lambda_min, 0) with bottom gaps
//      lambda_2 - lambda_1 = 0.04 .. 1.7 (scripts/dump_hessians.py), i.e. the rest of B's spectrum starts 5 .. 170 times
//      higher, and the iteration count is set by that single eigenvalue.  The Ritz step already holds its eigenvector u
//      (residual <= 1e-8 gap) and the filter's norm history gives a LOWER bound of the gap for free (ritz_eval), so the
//      iteration runs on  B~ = B + (tau - 1e-2) u u^T  (spectrum in [1e-2 + gap, s]: 2 iterations = 4 launches fewer on
//      closed-loop and on the bench's Hessians) and  B^(-1/2) = B~^(-1/2) + ((1e-2)^(-1/2) - tau^(-1/2)) u u^T  is put back
//      where Z is consumed (finalize launch, CovDeferred).  Not taken when the bound is small or the residual is not.
//   3. one workgroup: ONE Cholesky of Z (every update on the matrix cores, chol_lds.hpp) gives both
//      log det B and, scaled by sqrt(c/sqrt(s)), the factor L of Sigma.
// All reductions go through per-workgroup slots summed in a fixed order: results are bit-reproducible.
// Agreement with the LAPACK-eigh oracle: ~1e-13 relative (fp64), tests/test_gpu_parity.py.
// `batch` matrices per launch (grid.z / grid.y): covo-offline's 300-step table, env-batched configs.
#include "covo_common.hpp"
#include <cstdlib>
#include <cstring>
#include "wave_reduce.hpp"
#include "chol_lds.hpp"
#include "eps_tiles.hpp"
#include "sym_stats.hpp"
#include "noise_gemm_body.hpp"

typedef double f64x4 __attribute__((ext_vector_type(4)));

// -DNS_STAMPS: workgroup 0 of the iteration tail records wall_clock64() (10 ns) at the seams of every phase (scripts/ns_tail_cost.py)
#ifdef NS_STAMPS
__shared__ long long g_stamp[192];
__shared__ int g_nstamp;
#define NS_STAMP()                                                                    \
    do {                                                                              \
        if (threadIdx.x == 0 && g_nstamp < 192) g_stamp[g_nstamp++] = wall_clock64(); \
    } while (0)
#else
#define NS_STAMP() do { } while (0)
#endif

// -DNS_EVAL_STAMPS: wall_clock64() (10 ns ticks) of the squaring launch's seams into SC_STAMPS + 96 ..: [0] chain workgroup 0 enters,
// [1] leaves; per k = 2 .. 16 at [2 + 3 (k - 2)]: X_k seen complete, evaluated, decided; [70 ..]: the seams inside the evaluation of X_8 (scripts/ritz_timeline.py)
#ifdef NS_EVAL_STAMPS
#define EV_STAMP(s, i) do { if (threadIdx.x == 0) (s)[SC_STAMPS + 96 + (i)] = (double)wall_clock64(); } while (0)
#else
#define EV_STAMP(s, i) do { } while (0)
#endif
#define EV_STAMP8(s, k, i) do { if ((k) == 8) EV_STAMP(s, 70 + (i)); } while (0)
constexpr int SN = COVO_NA;  // 128
constexpr int NS_SQUARINGS = 16;   // cap: Chebyshev degree 2^16.  Real CoVO Hessians stop at 6..12; the cap matters for bottoms
                                   // that sit within 1e-5 of the spectrum's width of the next eigenvalues (round 3's fuzz sweep:
                                   // lambda_min = -0.015 next to the exact zeros of the null block, width 1.7e3: 15 squarings);
                                   // squarings beyond the 7th run inside the persistent tail launch and cost nothing once
                                   // stationary.  RITZ = 4 near-degenerate bottom eigenvalues are resolved exactly by the Ritz step
constexpr double NS_CUT_MARGIN = 1.0 / 1024.0;  // cut = min diag + margin * (hi - min diag): lambda_min is strictly amplified
constexpr double NS_SQ_TOL = 1e-7;     // squaring k+1 is skipped once |X_k|_F^2 moved by < 1e-7 relative ...
constexpr double NS_SQ_TGUARD = 1e10;  // ... and t_k > 1e10 (the bounded part of the spectrum is down at 1e-10)
constexpr int NS_ITERS = 12;       // scaled iteration: 8 for s/1e-2 = 1e3 (real CoVO Hessians), 10 for 2e4, 12 for 1e6; launches after
                                   // convergence return at once (1.6 us each)
constexpr double NS_TOL2 = 1e-8;   // iteration k+1 is skipped once |I - Z_k Y_k|_F^2 < 1e-8: step k itself squares that residual
                                   // (0.75 e^2 < 1e-8 in the Frobenius norm, ~4e-9 relative on Sigma -- 1/15 of an fp32 ulp; round 1
                                   // asked for 1e-10 here, i.e. 1e-20 after the last step, one iteration more on 1 step in 6)
constexpr int RITZ = 4;            // Rayleigh-Ritz block: exact lambda_min for up to 4 near-degenerate bottom eigenvalues

// per-matrix scratch (doubles): scalars, the Newton-Schulz coefficient table, per-row data of A, and the
// per-workgroup reduction slots (no atomics anywhere: fixed summation order, bit-reproducible)
enum { SC_SHIFT = 0, SC_LMIN, SC_DELTA, SC_SCALE, SC_SUMLOGB, SC_ZBUF, SC_ITERS, SC_KWIN, SC_SQ, SC_SQ_DONE, SC_NS_DONE,
       SC_FRO2, SC_TRACE, SC_GERSH, SC_N0,
       // (SC_SUMLOGB = 4: sum_i log diag(chol(B)), B = A + delta I: log det B = 2 SC_SUMLOGB -- ns_logdetB_workgroup)
       // (SC_KWIN = 7: the squaring count k whose Rayleigh-Ritz evaluation the chain took lambda_min from -- ritz_decide;
       //  SC_SQ: squarings started, which the early evaluations let run ahead of SC_KWIN)
       SC_PROF = 16,           // clock64() stamps of the finalize kernel (debug; 16-20), the persistent launches' modes (21, 22)
       SC_LDFLAG = 23,         // != 0: SC_SUMLOGB is there (zeroed with the scalars at the head of the chain)
       SC_BAR = 24,            // grid-barrier counters of the two persistent launches (unsigned in slots 24, 25; zeroed with the scalars)
       SC_BARFAIL = 26,        // != 0: a grid barrier timed out -> the finalize launch poisons Sigma and L with NaN
       SC_CZ = 27,             // Sigma = cz sym(Z): read by the noise GEMM when it writes a_cov for the finalize launch (CovDeferred)
       SC_LO = 28,             // lower end of the spectrum the Newton-Schulz table is built for: 1e-2, or 1e-2 + gap bound (deflation)
       SC_GAM = 29,            // deflation: Y0 = (B + (tau - 1e-2) u u^T) / s = B/s + gam u u^T (0: off)
       SC_ZCOEF = 30,          // deflation: Z = Z~ + zcoef u u^T, zcoef = sqrt(s) ((1e-2)^(-1/2) - tau^(-1/2)) (0: off)
       SC_GAPEST = 31,         // the gap bound lambda_2 - lambda_1 >= ... (diagnostics)
       SC_RESID = 15,          // |A u - theta u|^2 of the bottom Ritz pair (diagnostics)
       SC_COEF = 32,           // a_k, b_k   (2 * NS_ITERS)
       SC_ALPHA = 56, SC_BETA = 57,  // the filter's affine map Y0f = alpha I - beta A (first squaring; ns_first_setup rebuilds A^2 from X_1)
       SC_ROWABS = 64,         // sum_c |A[r][c]|            (128)
       SC_DIAG = 192,          // A[r][r]                    (128)
       SC_PREP = 320,          // prep partials: 8 x {max rowabs, sum v^2, trace, min diag}
       // the Rayleigh-Ritz evaluations of X_2 .. X_16 (slot + k - 2; all zeroed by the first squaring):
       SC_VERD = 320,          // verdicts: 0 not yet, 1 not taken, 2 taken (ritz_decide)
       SC_READY = 335,         // != 0: the chain's result (ritz_publish: U, delta, scale, lo, ...) is complete in memory -- what the
                               // Newton-Schulz workgroups of the MERGED launch wait for (round 6; cleared a launch ahead: KD / ns_prep_kernel)
       SC_HAVE_DIAG = 336,     // != 0: X_k's evaluation has read the diagonal of X_(k-1) (its column picks) ...
       SC_HAVE_COLS = 352,     // ... and its columns of X_k: the chain may come round to those buffers again (ns_square_evaluator)
       SC_SQ_FINAL = 368,      // != 0: the filter stopped by itself (stationary / cap) and X_(SC_SQ_FINAL) is its last iterate
       SC_SQN = 384,           // |X_i|_F^2 partials: (NS_SQUARINGS + 1) x 64 (36 used)
       SC_ERR = SC_SQN + (NS_SQUARINGS + 1) * 64,  // |Z_k Y_k - I|_F^2 partials: NS_ITERS x 64
       SC_RPART = SC_ERR + NS_ITERS * 64,          // sym_stats.hpp: per-row, per-column-block |.|-sums (128 x 8)
       SC_FPART = SC_RPART + 128 * 8,              // per-tile sums of squares (36)
       SC_U = SC_FPART + 64,                       // the bottom Ritz vector u (128)
       SC_FLAGS = SC_U + 128,                      // barrier flag words of the two persistent launches: 2 x 64 unsigned (ns_flag_barrier)
       SC_STAMPS = SC_FLAGS + 64,                  // -DNS_STAMPS: 192 time stamps of the iteration tail's workgroup 0
       SC_COUNT = SC_STAMPS + 192 };
constexpr double NS_DEFL_SAFETY = 0.7;   // the NS table assumes 1e-2 + 0.7 x (gap bound): the bound's linearisation is good to ~3 %
constexpr double NS_DEFL_MIN_GAP = 2e-2; // below this the interval shrinks by < 3x: not worth a rank-1 correction
constexpr double NS_DEFL_RESID = 1e-8;   // |A u - theta u| <= 1e-8 gap: eigenvector angle <= 1e-8, Sigma error <= 1e-7 relative
static_assert(SC_COEF + 2 * NS_ITERS <= SC_ROWABS, "scalar slots");

// ---- one 256-thread workgroup = one 16x16 tile of C = At^T . B  (At, B row-major 128x128); wave q takes
// the K-quarter [32q, 32q+32) (8 MFMAs), the four partial tiles are summed through LDS.
// MFMA f64 16x16x4: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; C/D: col = l&15, row = (l>>4) + 4*reg.
// The A operand is fetched as At[k][i], so BOTH operands are read as 128-B contiguous runs; callers pass
// the stored TRANSPOSE of the left factor.  `f(value, row, col)` maps the stored element to the operand
// (identity; alpha I - beta A for the first squaring).
struct LoadPlain {
    __device__ __forceinline__ double operator()(double v, int, int) const { return v; }
};
struct LoadAffine {  // alpha I - beta A
    double alpha, beta;
    __device__ __forceinline__ double operator()(double v, int r, int c) const { return fma(-beta, v, (r == c) ? alpha : 0.0); }
};
// COH = 1: the access is an agent-scope relaxed atomic (sc1): coherent across the XCDs' L2s without cache-wide fences.  Used by
// the persistent launches below, whose phases exchange tiles and slots across workgroups INSIDE one kernel; the same
// bodies compiled with COH = 0 (plain cached accesses) serve the one-launch-per-phase kernels.  Same arithmetic.
// COH = 2 (round 4): the persistent launch has VERIFIED that all its workgroups sit on ONE XCD (ns_flag_barrier).  Stores are then
// plain -- write-through the CU's L1 into the XCD's L2, which every reader shares -- and only the loads stay sc1 (they skip the
// L1s; a line that is dirty in the local L2 is served from there).  A phase drops from 2.9-3.3 to ~1.9 us
// (scripts/probe/xcd_chain_probe.hip: "sc1 loads, flag words polled sc1" against "sc1 loads + sc1 stores").
constexpr int COH_NONE = 0, COH_AGENT = 1, COH_XCD = 2;
template <int COH>
__device__ __forceinline__ double gld(const double *p)
{
    if (COH != COH_NONE) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <int COH>
__device__ __forceinline__ void gst(double *p, double v)
{
    if (COH == COH_AGENT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

struct TileOps {
    double a[8], b[8];
};
// issue the 16 operand loads of this wave's K-quarter (callers issue them BEFORE looking at any flag: every
// launch of the chain then pays one memory latency, not one per dependent scalar)
template <int COH = COH_NONE, class F>
__device__ __forceinline__ void tile_load(TileOps &o, const double *A, const double *B, int ti, int tj, int lane, int kq, F f)
{
    const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int k = 32 * kq + 4 * kk + hi;
        o.a[kk] = f(gld<COH>(A + (size_t)k * SN + 16 * ti + lo), k, 16 * ti + lo);
        o.b[kk] = f(gld<COH>(B + (size_t)k * SN + 16 * tj + lo), k, 16 * tj + lo);
    }
}
__device__ __forceinline__ f64x4 tile_mma(const TileOps &o)
{
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[kk], o.b[kk], acc, 0, 0, 0);
    return acc;
}
// sums the four K-quarters; wave wv returns the tile element (row, col) = ((lane>>4) + 4 wv, lane&15) it will store
__device__ __forceinline__ double tile_reduce(const f64x4 &acc, double (*red)[4][64], int wv, int lane)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv][r][lane] = acc[r];
    __syncthreads();
    return (red[0][wv][lane] + red[1][wv][lane]) + (red[2][wv][lane] + red[3][wv][lane]);
}
// workgroup sum (256 threads) of one double per lane, fixed order; valid in every thread
__device__ __forceinline__ double wg_sum4(double v, double *part, int wv, int lane)
{
    v = wr::wave64_allsum(v);
    if (lane == 0) part[wv] = v;
    __syncthreads();
    return (part[0] + part[1]) + (part[2] + part[3]);
}
// sum of n <= 64 slots, fixed order; valid in every lane
__device__ __forceinline__ double slot_sum(const double *__restrict__ p, int n, int lane)
{
    return wr::wave64_allsum((lane < n) ? p[lane] : 0.0);
}

// (matrix, tile) of this workgroup.  Batch 1: grid (tiles).  Batched: grid (tiles, batch rounded up to 8); the linear workgroup id
// round-robins over the 8 XCDs, so matrix b takes the ids = b (mod 8): ALL the tiles of one matrix run on one XCD and every
// launch of the chain finds its operands (128 KB each, written by the launch before on the same XCD) behind ONE L2 -- with
// (blockIdx.x, blockIdx.y) = (tile, matrix) every XCD pulled every matrix through the fabric.  Same tiles, same arithmetic.
__device__ __forceinline__ bool ns_block(int batch, int &b, int &w)
{
    if (gridDim.y == 1) {
        b = 0;
        w = (int)blockIdx.x;
        return true;
    }
    const unsigned id = blockIdx.y * gridDim.x + blockIdx.x, slot = id >> 3;
    b = (int)(slot / gridDim.x) * 8 + (int)(id & 7u);
    w = (int)(slot % gridDim.x);
    return b < batch;
}
static inline dim3 ns_grid(int tiles, int batch) { return dim3(tiles, batch == 1 ? 1 : (batch + 7) / 8 * 8); }

// The squaring stage (a filter iteration: self-correcting) works on exactly symmetric matrices: only tiles
// with ti >= tj are computed and every value is stored together with its mirror image (36 tiles).
constexpr int NS_TILES = 36;
__device__ __forceinline__ void tri_tile(int w, int &ti, int &tj)
{
    ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= w) ++ti;
    tj = w - ti * (ti + 1) / 2;
}
// (staging the transposed copy through LDS so that both orientations leave as 128-byte row segments was measured and
// changes nothing, neither behind the write-back L2 nor with the write-through stores of the persistent launches)
template <int COH = COH_NONE>
__device__ __forceinline__ void store_both(double *O, double *Ot, int row, int col, double v)
{
    gst<COH>(O + (size_t)row * SN + col, v);
    gst<COH>(Ot + (size_t)col * SN + row, v);
}
template <int COH = COH_NONE>
__device__ __forceinline__ void store_sym(double *O, int row, int col, double v)
{
    if (row >= col) {
        gst<COH>(O + (size_t)row * SN + col, v);
        if (row > col) gst<COH>(O + (size_t)col * SN + row, v);
    }
}

// ---- prep (one workgroup per lower 16x16 tile): A = (R + R^T)/2 (covo.py:117, bitwise symmetric) and the tile's share of the
// input statistics (sym_stats.hpp).  The affine map Y0 = alpha I - beta A is formed by the first squaring on load.  The fused
// step does not run this launch: its Hessian is exactly symmetric and KD leaves the same statistics (A is then R itself).
__device__ __forceinline__ SymStatsOut ns_stats_out(double *s)
{
    SymStatsOut o;
    o.rpart = s + SC_RPART;
    o.fpart = s + SC_FPART;
    o.diag = s + SC_DIAG;
    o.stride = 0;
    o.flags = nullptr;
    o.ready = nullptr;
    return o;
}
__global__ __launch_bounds__(256) void ns_prep_kernel(const double *__restrict__ Rin, double *__restrict__ Aall,
                                                      double *__restrict__ scall, int batch)
{
    __shared__ double tmp[16][17];
    __shared__ double part[4];
    int b, w;
    if (!ns_block(batch, b, w)) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double *R = Rin + (size_t)b * SN * SN;
    double *A = Aall + (size_t)b * SN * SN;
    double *s = scall + (size_t)b * SC_COUNT;
    if (w == 0 && tid < SC_COEF) s[tid] = 0.0;  // scalars and the two "done" flags
    if (w == 0 && tid < 64) s[SC_FLAGS + tid] = 0.0;
    if (w == 0 && tid == 64) s[SC_READY] = 0.0;  // (a launch ahead of the merged launch whose Newton-Schulz workgroups poll it)
    int I, J;
    tri_tile(w, I, J);
    const int i = 16 * I + (lane >> 4) + 4 * wv, j = 16 * J + (lane & 15);
    const double v = 0.5 * (R[(size_t)i * SN + j] + R[(size_t)j * SN + i]);
    A[(size_t)i * SN + j] = v;
    A[(size_t)j * SN + i] = v;
    sym_tile_stats(true, v, I, J, w, lane, wv, ns_stats_out(s), tmp, part);
}

// the first squaring's prologue: the input statistics from the per-tile partials (sym_stats.hpp), every workgroup for itself in the
// same fixed order -> alpha, beta of the affine map Y0 = alpha I - beta A and |Y0|_F^2; workgroup 0 keeps the per-row sums for the
// Ritz launch and clears the scalars (inside the persistent launch coherently -- the "done" flags are polled by every workgroup
// from the second squaring on; the barrier flag words have been cleared a launch earlier: KD or ns_prep_kernel).
// scr: >= 12 doubles of LDS; two barriers inside (scr is free again on return).
template <int COH>
__device__ __forceinline__ void ns_square_first_stats(double *s, int w, int tid, int lane, int wv, double *scr, double &alpha,
                                                      double &beta, double &nrm)
{
    if (w == 0 && tid < SC_COEF) gst<COH>(s + tid, 0.0);
    if (w == 0 && tid < 49 && SC_VERD + tid != SC_READY) gst<COH>(s + SC_VERD + tid, 0.0);  // the evaluations' slots and SC_SQ_FINAL
    if (COH == COH_NONE && w == 0 && tid < 64) s[SC_FLAGS + tid] = 0.0;
    double ra = 0.0, dgv = 0.0;
    if (tid < SN) {
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) ra += s[SC_RPART + tid * 8 + cb];
        dgv = s[SC_DIAG + tid];
        if (w == 0) gst<COH>(s + SC_ROWABS + tid, ra);  // (read by the evaluating workgroups of the same launch: ritz_eval)
    }
    const double fp = (lane < NS_TILES) ? s[SC_FPART + lane] : 0.0;
    const double f2 = wr::wave64_allsum(fp);
    const double gmw = wr::wave64_allmax(tid < SN ? ra : 0.0), mdw = -wr::wave64_allmax(tid < SN ? -dgv : -1e300);
    const double trw = wr::wave64_allsum(tid < SN ? dgv : 0.0);
    if (lane == 0) {
        scr[wv] = gmw;
        scr[4 + wv] = mdw;
        scr[8 + wv] = trw;
    }
    __syncthreads();
    const double gm = fmax(scr[0], scr[1]), md = fmin(scr[4], scr[5]), tr = scr[8] + scr[9];
    __syncthreads();  // scr is reused by the caller
    // any hi >= lambda_max(A) and any cut > lambda_min(A) work; the tighter they are the faster the filter separates
    const double hi = fmin(gm, sqrt(f2)) * (1.0 + 1e-12) + 1e-3;
    const double cut = fma(NS_CUT_MARGIN, hi - md, md);
    const double inv = 1.0 / (hi - cut);
    alpha = (hi + cut) * inv;
    beta = 2.0 * inv;
    nrm = fma((double)SN * alpha, alpha, fma(-2.0 * alpha * beta, tr, beta * beta * f2));  // |alpha I - beta A|_F^2
    if (w == 0 && tid == 0) {
        gst<COH>(s + SC_SHIFT, hi);
        gst<COH>(s + SC_FRO2, f2);
        gst<COH>(s + SC_TRACE, tr);
        gst<COH>(s + SC_GERSH, gm);
        gst<COH>(s + SC_N0, nrm);
        gst<COH>(s + SC_ALPHA, alpha);
        gst<COH>(s + SC_BETA, beta);
    }
}

// ---- one doubling of the Chebyshev degree: Xout = Xin^2 / |Xin|_F^2 - I / t_out, t_out = 2 t_in^2 |Xin|_F^2 (36 lower
// tiles); |Xout|_F^2 partials go to slot row step+1, t_out to slot 63 of that row.
// FIRST: Xin is A and the operand is Y0 = alpha I - beta A (see the header), t_in = 1.
// Returns false when the filter is found stationary (nothing was written).  Workgroup w of matrix b.
template <bool FIRST, int COH>
__device__ __forceinline__ bool ns_square_body(const double *Xin, double *Xout, double *scall, int step, int b, int w,
                                               double (*red)[4][64], double *part)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double *X = Xin + (size_t)b * SN * SN;
    double *O = Xout + (size_t)b * SN * SN;
    double *s = scall + (size_t)b * SC_COUNT;
    double nrm, t_in = 1.0, alpha = 0.0, beta = 0.0;
    if (FIRST) ns_square_first_stats<COH>(s, w, tid, lane, wv, &red[0][0][0], alpha, beta, nrm);
    int ti, tj;
    tri_tile(w, ti, tj);
    // batched launches (gridDim.y > 1: the env-batched step, covo-offline's table): every squaring the cap allows is launched for
    // every matrix, and most matrices are done long before the slowest -- a finished one must not pull its 32 KB of operands
    // through the L2s again (round 4: the flag first there; batch 1 keeps the operand loads ahead of the flag, one round trip)
    if (!FIRST && !COH && gridDim.y > 1 && s[SC_SQ_DONE] != 0.0) return false;
    TileOps ops;
    if (FIRST) tile_load<COH>(ops, X, X, ti, tj, lane, wv, LoadAffine{alpha, beta});
    else tile_load<COH>(ops, X, X, ti, tj, lane, wv, LoadPlain{});
    if (!FIRST) {
        const double done = gld<COH>(s + SC_SQ_DONE);
        const double p1 = (lane < NS_TILES) ? gld<COH>(s + SC_SQN + step * 64 + lane) : 0.0;
        const double p0 = (lane < NS_TILES && step >= 2) ? gld<COH>(s + SC_SQN + (step - 1) * 64 + lane) : 0.0;
        t_in = gld<COH>(s + SC_SQN + step * 64 + 63);
        if (done != 0.0) return false;
        nrm = wr::wave64_allsum(p1);
        if (step >= 2 && t_in > NS_SQ_TGUARD) {
            // stationary: the bounded part of the spectrum is gone and |X_k|_F^2 stopped moving (X is a projector
            // onto the bottom eigenspace up to scale)
            const double prev = wr::wave64_allsum(p0);
            if (fabs(nrm - prev) <= NS_SQ_TOL * nrm) {
                if (w == 0 && tid == 0) {
                    gst<COH>(s + SC_SQ_FINAL, (double)step);  // X_step is the filter's last iterate
                    gst<COH>(s + SC_SQ_DONE, 1.0);
                }
                return false;
            }
        }
    }
    const double t_out = 2.0 * t_in * t_in * nrm;  // overflows to +inf once the filter has separated: 1 / t_out = 0
    const double inv_t = 1.0 / t_out;
    if (w == 0 && tid == 0) {
        gst<COH>(s + SC_SQ, (double)(step + 1));
        gst<COH>(s + SC_SQN + (step + 1) * 64 + 63, t_out);
    }
    const f64x4 acc = tile_mma(ops);
    const int row = 16 * ti + (lane >> 4) + 4 * wv, col = 16 * tj + (lane & 15);
    const double v = tile_reduce(acc, red, wv, lane) * (1.0 / nrm) - ((row == col) ? inv_t : 0.0);
    store_sym<COH>(O, row, col, v);
    const double tot = wg_sum4((row > col) ? 2.0 * v * v : ((row == col) ? v * v : 0.0), part, wv, lane);
    if (tid == 0) gst<COH>(s + SC_SQN + (step + 1) * 64 + w, tot);
    return true;
}

template <bool FIRST>
__global__ __launch_bounds__(256) void ns_square_kernel(const double *__restrict__ Xin, double *__restrict__ Xout,
                                                        double *__restrict__ scall, int step, int batch)
{
    __shared__ double red[4][4][64];
    __shared__ double part[4];
    int b, w;
    if (!ns_block(batch, b, w)) return;
    (void)ns_square_body<FIRST, COH_NONE>(Xin, Xout, scall, step, b, w, red, part);
}

// ---- Rayleigh-Ritz on the RITZ largest-diagonal columns of X_k, the filter after k squarings: lambda_min(A), delta; then the
// scale s of B = A + delta I (min of its Gershgorin and Frobenius bounds, from the per-row data of prep) and the deflation data.
//
// Round 5 -- WHICH X_k.  Rounds 1-4 ran this once, on the iterate at which the filter's norm had become stationary (|X_k|_F^2 moving
// by < 1e-7: the second eigenvalue suppressed to ~5e-8).  But a RITZ-column block recovers the bottom eigenvector as soon as the
// FIFTH eigenvalue is suppressed, and the Ritz pair says so itself: its residual |A u - theta u| is a guaranteed error bound.  On the
// Hessians of closed-loop episodes (scripts/analysis/filter_emul.py on scripts/dump_hessians.py's dump) the pair of X_k passes the
// deflation's own acceptance test -- residual <= 1e-8 x (0.7 x the gap bound), gap bound > 2e-2 -- 2.5 squarings before the norm
// goes stationary (k = 8.6 against 11.1; |lambda error| <= 5e-14 at that k).  So the chain's lambda_min is DEFINED as
//     the Ritz value of X_kwin,  kwin = the first k in [2, 16] whose pair passes (ritz_eval: pass), else the filter's last iterate,
// a pure function of A: every path evaluates the same X_k with the same code (ritz_eval) and takes the same kwin (ritz_decide) --
//   * one matrix, persistent launch: evaluating workgroups ride in the squaring launch (ns_square_tail_pair_kernel<true>), read X_k
//     the moment its barrier has passed, and stop the chain; the chain meanwhile runs ahead (every X_k keeps a buffer of its own),
//     so an evaluation costs the chain nothing and the separate Ritz launch is gone;
//   * batches / shared-device handles / the debug splits: the squarings run to their own stop as before and ONE launch evaluates
//     every k of every matrix (ns_ritz_scan_kernel).
constexpr double RITZ_PASS_E = 0.05;  // e_k = (1 - |X_k|_F^2) / 2 below this: the gap bound's linearisation holds (ritz_eval)
constexpr int RITZ_K0 = 2;                          // first evaluated iterate (the stationarity test needs two norms: the filter never stops before X_2)
constexpr int RITZ_NK = NS_SQUARINGS - RITZ_K0 + 1;  // evaluations per matrix: X_2 .. X_16
struct XBufs {  // X_1 -> xq (it outlives the filter: iteration 1 of the Newton-Schulz part rebuilds A^2 from it, NsFirst); X_2 -> x1, and then
    double *x0, *x1, *hist, *xq;  // X_3 .. X_16 -> the history;  hist == nullptr: no history, odd k -> x0, even k -> x1 (the launch with the
    size_t M;                     // evaluations inside: they take what they need of X_k before the chain comes round to its buffer
                                  // again -- ns_square_evaluator).  x0, x1 are T and T^T of the iterations later.
                                  // M: doubles between the buffers of consecutive k (batch x 128 x 128)
};
__host__ __device__ __forceinline__ double *ns_xk(const XBufs xb, int k)
{
    if (k == 1) return xb.xq;
    if (xb.hist == nullptr) return (k & 1) ? xb.x0 : xb.x1;
    return (k == 2) ? xb.x1 : xb.hist + (size_t)(k - 3) * xb.M;
}
struct RitzLds {
    double V[RITZ][SN];
    double AVp[4][RITZ][SN];  // partial A V over the four column quarters
    double H[RITZ][RITZ];
    double red[2];
    double nrm[NS_SQUARINGS + 1];  // |X_j|_F^2 of every filter step
    double cvec[RITZ], lmin, gap, md[2], r2[2];
    int has_null, null_is_min;
    // the evaluation's results (ritz_publish)
    double o_lmin, o_scale, o_lo, o_gam, o_zc, o_gapest, o_resid2;
    int o_pass;
    int decide, abort, prev_hint;  // prev_hint: the verdict on X_(k-1) as ritz_eval saw it (0: not yet)
    int pick[RITZ];
    double C[RITZ][SN];            // the picked columns of X_k as loaded (the dominance test: ritz_eval)
    double numq[2][RITZ], uq[RITZ];
};
// what an evaluating workgroup keeps of A and of the chain's input statistics (loaded once per matrix; 256 threads)
template <bool LEAN = false>
struct RitzInT {
    double acol[2][LEAN ? 1 : 32];  // thread (r = tid & 63, g = tid >> 6): A[32 g .. 32 g + 31][r] and [..][r + 64] -- two rows share every V they read
    double dg, ra;    // tid < 128: A[tid][tid], sum_c |A[tid][c]|
    double tr, f2, hi;
    const double *A;  // LEAN evaluations (the batched squaring launch: three workgroups per CU, <= 168 VGPRs) read A where it lies
};
typedef RitzInT<false> RitzIn;
// (two parts: A and its diagonal are there when the launch starts; the row sums and bounds are left by the first squaring)
template <bool LEAN = false>
__device__ __forceinline__ void ritz_load_matrix(RitzInT<LEAN> &in, const double *__restrict__ A, const double *s, int tid)
{
    const int r_av = tid & 63, g = tid >> 6;
    in.A = A;
    if (!LEAN) {
#pragma unroll
        for (int c = 0; c < 32; ++c) {  // A symmetric: coalesced in r
            in.acol[0][c] = A[(size_t)(32 * g + c) * SN + r_av];
            in.acol[1][c] = A[(size_t)(32 * g + c) * SN + r_av + 64];
        }
    }
    in.dg = (tid < SN) ? s[SC_DIAG + tid] : 0.0;  // (written a launch earlier: KD / ns_prep_kernel)
}
template <int COH, bool LEAN = false>
__device__ __forceinline__ void ritz_load_inputs(RitzInT<LEAN> &in, const double *s, int tid)
{
    in.ra = (tid < SN) ? gld<COH>(s + SC_ROWABS + tid) : 0.0;
    in.tr = gld<COH>(s + SC_TRACE);
    in.f2 = gld<COH>(s + SC_FRO2);
    in.hi = gld<COH>(s + SC_SHIFT);
}
// One evaluation (256 threads; barriers inside): the bottom Ritz pair of X = X_k and everything the iterations need from it, left
// in L.o_*; returns this thread's component of u (tid < 128).  COH: how X and the norm slots are read (the chain may still be running).
// ABORT (the evaluations inside the squaring launch): every seam looks at SC_KWIN -- once another evaluation has been taken this
// one is void and the workgroup must leave (the launch ends when its last workgroup does): returns with L.abort != 0.
// The RITZ columns are PICKED on the diagonal of the iterate BEFORE (X_(k-1): the largest entries; the same eigenvectors, weights one
// squaring younger): an evaluating workgroup of the squaring launch does this while X_k is still being computed, and X_k then costs
// one round trip (the four columns), not two.  Wave 0 only; leaves L.pick[] and L.has_null.
template <int COH>
__device__ __forceinline__ void ritz_picks(const double *__restrict__ Xp, const double *s, RitzLds &L)
{
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid >= 64) return;
    double d0 = gld<COH>(Xp + (size_t)lane * SN + lane), d1 = gld<COH>(Xp + (size_t)(lane + 64) * SN + lane + 64);
    // Rows of A that are exactly zero (always the last four of a CoVO Hessian: a_(H-1) never reaches a reward) are unit
    // eigenvectors of eigenvalue 0.  While the filter has not yet separated lambda_min < 0 from them they own the largest
    // diagonal entries of X and would take all RITZ picks (round 3's fuzz sweep: lambda_min = -0.015 reported as 0, B
    // indefinite, NaN): they are left out of the picks and enter as the known eigenvalue 0 below.
    const bool null0 = gld<COH>(s + SC_ROWABS + lane) == 0.0, null1 = gld<COH>(s + SC_ROWABS + lane + 64) == 0.0;
    const int n_null = __builtin_popcountll(__ballot(null0)) + __builtin_popcountll(__ballot(null1));
    const bool skip_null = n_null > 0 && n_null <= SN - RITZ;
    if (skip_null) {
        if (null0) d0 = -1e300;
        if (null1) d1 = -1e300;
    }
    if (lane == 0) L.has_null = skip_null ? 1 : 0;
#pragma unroll
    for (int q = 0; q < RITZ; ++q) {
        // argmax over 128 values: wave maximum (DPP), then the first lane/slot holding it
        const double mx = wr::wave64_allmax(fmax(d0, d1));
        const unsigned long long m0 = __ballot(d0 == mx), m1 = __ballot(d1 == mx);
        const int bi = m0 ? (int)__builtin_ctzll(m0) : 64 + (int)__builtin_ctzll(m1);
        if (lane == 0) L.pick[q] = bi;
        if (bi == lane) d0 = -1e300;
        if (bi == lane + 64) d1 = -1e300;
    }
}
// One evaluation (256 threads; barriers inside): the bottom Ritz pair of X = X_k on the columns L.pick[] (ritz_picks on X_(k-1),
// same workgroup) and everything the iterations need from it, left in L.o_*; returns this thread's component of u (tid < 128).
// COH: how X and the norm slots are read (the chain may still be running).
// ABORT (the evaluations inside the squaring launch): every seam looks at SC_KWIN -- once another evaluation has been taken this
// one is void and the workgroup must leave (the launch ends when its last workgroup does): returns with L.abort != 0.
// have_cols != nullptr: raised (coherently) once the columns of X_k are in registers -- the chain may then reuse X_k's buffer.
template <int COH, bool ABORT = false, bool LEAN = false>
__device__ __forceinline__ double ritz_eval(const RitzInT<LEAN> &in, const double *__restrict__ X, const double *s, int k, RitzLds &L,
                                            double *have_cols = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double kw = 0.0;  // (each seam's load is issued a phase ahead of its use)
    if (ABORT) kw = gld<COH_AGENT>(s + SC_KWIN);
    const double t_k = gld<COH>(s + SC_SQN + k * 64 + 63);  // (wanted at the very end)
    __syncthreads();  // (L is reused from the evaluation before; L.pick is there)
    if (ABORT && tid == 0) L.abort = 0;
    EV_STAMP8(const_cast<double *>(s), k, 0);
    if (tid < 64) {
        // ---- wave 0: the picked columns, orthonormalised (classical Gram-Schmidt, twice -- the dot products of a pass are
        // independent reductions; everything in registers: lane l owns rows l and l+64; reductions on the VALU)
        double v[RITZ][2];
        int pick_k[RITZ];
#pragma unroll
        for (int q = 0; q < RITZ; ++q) pick_k[q] = L.pick[q];
#pragma unroll
        for (int q = 0; q < RITZ; ++q) {
            v[q][0] = gld<COH>(X + (size_t)lane * SN + pick_k[q]);
            v[q][1] = gld<COH>(X + (size_t)(lane + 64) * SN + pick_k[q]);
        }
        if (have_cols != nullptr) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) gst<COH_AGENT>(have_cols, 1.0);
        }
#pragma unroll
        for (int q = 0; q < RITZ; ++q) {
            L.C[q][lane] = v[q][0];
            L.C[q][lane + 64] = v[q][1];
        }
#pragma unroll
        for (int q = 0; q < RITZ; ++q) {
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                double dot[RITZ];
#pragma unroll
                for (int j = 0; j < q; ++j) dot[j] = wr::wave64_allsum(fma(v[q][0], v[j][0], v[q][1] * v[j][1]));
#pragma unroll
                for (int j = 0; j < q; ++j) {
                    v[q][0] = fma(-dot[j], v[j][0], v[q][0]);
                    v[q][1] = fma(-dot[j], v[j][1], v[q][1]);
                }
            }
            const double n2 = wr::wave64_allsum(fma(v[q][0], v[q][0], v[q][1] * v[q][1]));
            if (n2 > 1e-280) {
                const double inv = qm::rsq64_(n2);
                v[q][0] *= inv;
                v[q][1] *= inv;
            } else {  // column numerically inside the span of the previous ones: any unit vector will do
                v[q][0] = (pick_k[q] == lane) ? 1.0 : 0.0;
                v[q][1] = (pick_k[q] == lane + 64) ? 1.0 : 0.0;
            }
            L.V[q][lane] = v[q][0];
            L.V[q][lane + 64] = v[q][1];
        }
    } else {
        // waves 1 .. 3 meanwhile: the filter's norm history (rows 1 .. k: 36 partials each, fixed order) for the gap bound below
        for (int j = wave; j <= NS_SQUARINGS; j += 3) {
            const double p = (j <= k && lane < NS_TILES) ? gld<COH>(s + SC_SQN + j * 64 + lane) : 0.0;
            const double vv = (j <= k) ? wr::wave64_allsum(p) : 1.0;
            if (lane == 0) L.nrm[j] = vv;
        }
        if (wave == 1) {
            const double m = -wr::wave64_allmax(-in.dg);  // (rows 64 .. 127; rows 0 .. 63 below)
            if (lane == 0) L.md[1] = m;
        }
    }
    if (wave == 0) {
        const double m = -wr::wave64_allmax(-in.dg);
        if (lane == 0) L.md[0] = m;
    }
    if (ABORT && tid == 0 && kw != 0.0) L.abort = 1;
    if (ABORT) kw = gld<COH_AGENT>(s + SC_KWIN);
    __syncthreads();
    EV_STAMP8(const_cast<double *>(s), k, 1);
    if (ABORT && L.abort) return 0.0;
    // A V: this thread's column quarter for its two rows (the LDS reads of V are what this phase waits for: each feeds two
    // fused multiply-adds), then H = V^T A V from the four partials
    {
        const int r_av = tid & 63, g = tid >> 6;
        double av[2][RITZ] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
        if (LEAN) {  // the same sums, A from memory in chunks of eight columns (the next chunk in flight)
            double a[2][2][8];
            // (the chunk's base address goes through an opaque asm: left alone the compiler forms all 64 addresses of all chunks
            // ahead of the workgroup's evaluation loop and spills them)
            // (... the OFFSET, not the pointer: a pointer that has been through an asm is generic to the compiler and its loads become
            // flat_load, which also count on lgkmcnt -- covo_common.hpp: rebase_global)
            size_t aoff = (size_t)(32 * g) * SN + r_av;
            asm volatile("" : "+v"(aoff));
            const double *Ap = in.A + aoff;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                a[0][0][c] = Ap[(size_t)c * SN];
                a[0][1][c] = Ap[(size_t)c * SN + 64];
            }
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                if (ch < 3) {
                    size_t noff = aoff + (size_t)(8 * (ch + 1)) * SN;
                    asm volatile("" : "+v"(noff));
                    const double *An = in.A + noff;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        a[(ch + 1) & 1][0][c] = An[(size_t)c * SN];
                        a[(ch + 1) & 1][1][c] = An[(size_t)c * SN + 64];
                    }
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
#pragma unroll
                    for (int q = 0; q < RITZ; ++q) {
                        const double vq = L.V[q][32 * g + 8 * ch + c];
                        av[0][q] = fma(a[ch & 1][0][c], vq, av[0][q]);
                        av[1][q] = fma(a[ch & 1][1][c], vq, av[1][q]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 32; ++c) {
#pragma unroll
                for (int q = 0; q < RITZ; ++q) {
                    const double vq = L.V[q][32 * g + c];
                    av[0][q] = fma(in.acol[0][c], vq, av[0][q]);
                    av[1][q] = fma(in.acol[1][c], vq, av[1][q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < RITZ; ++q) {
            L.AVp[g][q][r_av] = av[0][q];
            L.AVp[g][q][r_av + 64] = av[1][q];
        }
    }
    if (ABORT && tid == 0 && kw != 0.0) L.abort = 1;
    if (ABORT) kw = gld<COH_AGENT>(s + SC_KWIN);
    __syncthreads();
    EV_STAMP8(const_cast<double *>(s), k, 2);
    if (ABORT && L.abort) return 0.0;
#pragma unroll
    for (int e = 4 * wave; e < 4 * wave + 4; ++e) {  // 4 waves x 4 entries of H
        const int i = e / RITZ, j = e % RITZ;
        const double a0 = (L.AVp[0][j][lane] + L.AVp[1][j][lane]) + (L.AVp[2][j][lane] + L.AVp[3][j][lane]);
        const double a1 = (L.AVp[0][j][lane + 64] + L.AVp[1][j][lane + 64]) + (L.AVp[2][j][lane + 64] + L.AVp[3][j][lane + 64]);
        const double d = wr::wave64_allsum(fma(L.V[i][lane], a0, L.V[i][lane + 64] * a1));
        if (lane == 0) L.H[i][j] = d;
    }
    if (ABORT && tid == 0 && kw != 0.0) L.abort = 1;
    if (ABORT) kw = gld<COH_AGENT>(s + SC_KWIN);
    __syncthreads();
    EV_STAMP8(const_cast<double *>(s), k, 3);
    if (ABORT && L.abort) return 0.0;
    if (tid == 64) {
        // ---- a LOWER bound of the bottom gap lambda_2 - lambda_1 from the filter's norm history, on another wave while lane 0
        // diagonalises H.  X_k = sigma_k (v1 v1^T + sum_(i>=2) r_i v_i v_i^T), r_i = T_(2^k)(y_i) / T_(2^k)(y_1) (y = alpha -
        // beta lambda; eigenvalues inside [cut, hi] have |T| <= 1, negligible), and one squaring gives |X_(k+1)|_F^2 =
        // (1 + sum r_i^4) / (1 + sum r_i^2)^2 ~ 1 - 2 sum r_i^2.  With e = (1 - |X_j|_F^2) / 2 >= r_2^2 = exp(-2^j D),
        // D = acosh y_1 - acosh y_2:  D >= -ln(e) / 2^j, hence y_2 <= cosh(acosh y_1 - D_est) and lambda_2 >= (alpha - y_2) / beta.
        // lambda_1 enters through y_1 only to ~1 %: the Rayleigh quotient H[0][0] of the dominant column serves.
        double gap = 0.0;
        const double hi = in.hi, md = fmin(L.md[0], L.md[1]);
        const double cut = fma(NS_CUT_MARGIN, hi - md, md), inv = 1.0 / (hi - cut);
        const double alpha = (hi + cut) * inv, beta = 2.0 * inv, l1 = L.H[0][0];
        const double y1 = fma(-beta, l1, alpha);
        for (int j = 1; j <= k && j <= NS_SQUARINGS; ++j) {
            const double e = 0.5 * (1.0 - L.nrm[j]);
            if (e > 1e-13 && e < RITZ_PASS_E && y1 > 1.0) {  // the earliest step at which the linearisation holds
                const double D = -log(e) * exp2(-(double)j);
                const double ac = log(y1 + sqrt(fma(y1, y1, -1.0))) - D;
                const double ex = exp(fmax(ac, 0.0));
                const double y2 = 0.5 * (ex + 1.0 / ex);
                gap = fmax((alpha - y2) / beta - l1, 0.0);
                break;
            }
        }
        L.gap = gap;
    }
    if (tid < RITZ) {
        // Jacobi on the RITZ x RITZ symmetric H (exits when diagonal); square roots by hardware seed + Newton (qm::rsq64_), not libm.
        // Lanes 0..3 run it in lockstep, each with the whole of h and ONE row of the accumulated rotation (the eigenvectors for 4
        // extra operations per rotation instead of 16).  Round-robin order -- {(0,1),(2,3)}, {(0,2),(1,3)}, {(0,3),(1,2)}: the two
        // rotations of a round touch disjoint pivots, so their angles are ONE chain of dependent instructions (even lanes work out
        // the first pair's, odd lanes the second's; the results travel by v_readlane) -- the chain is what a sweep costs.
        double h[RITZ][RITZ], jrow[RITZ];
#pragma unroll
        for (int i = 0; i < RITZ; ++i) {
#pragma unroll
            for (int j = 0; j < RITZ; ++j) h[i][j] = 0.5 * (L.H[i][j] + L.H[j][i]);
            jrow[i] = (i == tid) ? 1.0 : 0.0;
        }
        const bool second = (tid & 1) != 0;
        auto bcast = [](double x, int l) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
            return __hiloint2double(hi, lo);
        };
        auto rotate = [&](int p, int q2, double c, double sn, double t) {
            // J^T h J for the symmetric h, written out: the 2 x 2 pivot block in closed form (h_pq -> 0), the other rows and
            // columns once, mirrored
            const double hpq = h[p][q2] * t;
            h[p][p] -= hpq;
            h[q2][q2] += hpq;
            h[p][q2] = 0.0;
            h[q2][p] = 0.0;
#pragma unroll
            for (int kk = 0; kk < RITZ; ++kk) {
                if (kk == p || kk == q2) continue;
                const double hp = h[kk][p], hq = h[kk][q2];
                const double np_ = c * hp - sn * hq, nq_ = sn * hp + c * hq;
                h[kk][p] = np_;
                h[p][kk] = np_;
                h[kk][q2] = nq_;
                h[q2][kk] = nq_;
            }
            const double jp = jrow[p], jq = jrow[q2];  // this lane's row of the accumulated rotation: columns p, q2
            jrow[p] = c * jp - sn * jq;
            jrow[q2] = sn * jp + c * jq;
        };
#pragma unroll 1
        for (int sweep = 0; sweep < 12; ++sweep) {  // (h and jrow stay in registers: every inner loop is unrolled)
            double off = 0.0, dia = 0.0;
#pragma unroll
            for (int p = 0; p < RITZ; ++p)
#pragma unroll
                for (int q2 = 0; q2 < RITZ; ++q2) (p == q2 ? dia : off) += h[p][q2] * h[p][q2];
            if (off <= 1e-26 * dia) break;  // off^2 / gap bounds the eigenvalue error: far below the 1e-13 the chain needs
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int p1 = 0, q1 = r + 1, p2 = (r == 0) ? 2 : 1, q2 = (r == 2) ? 2 : 3;
                const double hpp = second ? h[p2][p2] : h[p1][p1], hqq = second ? h[q2][q2] : h[q1][q1];
                const double hpq = second ? h[p2][q2] : h[p1][q1];
                const bool skip = hpq * hpq <= 1e-34 * fabs(hpp * hqq);  // (-> the identity: c = 1, s = t = 0)
                // the small rotation that zeroes h_pq: tan(2 theta) = b / a, a = h_qq - h_pp, b = 2 h_pq;  cos(2 theta) = |a| / r,
                // r = sqrt(a^2 + b^2);  c^2 = (1 + cos 2 theta) / 2;  s = sign(a) b / (2 r c);  t = s / c -- two rsqrt chains
                const double ja = skip ? 1.0 : hqq - hpp, jb = skip ? 0.0 : 2.0 * hpq;
                const double inv_r = qm::rsq64_(fma(ja, ja, jb * jb));
                const double c2 = fma(0.5 * fabs(ja), inv_r, 0.5);
                const double inv_c = qm::rsq64_(c2);
                const double c = c2 * inv_c;
                const double sn = copysign(0.5, ja) * jb * inv_r * inv_c;
                const double t = sn * inv_c;
                const double ca = bcast(c, 0), sa = bcast(sn, 0), ta = bcast(t, 0);
                const double cb = bcast(c, 1), sb = bcast(sn, 1), tb = bcast(t, 1);
                rotate(p1, q1, ca, sa, ta);
                rotate(p2, q2, cb, sb, tb);
            }
        }
        double lmin = h[0][0], cmin = jrow[0];
#pragma unroll
        for (int i = 1; i < RITZ; ++i)
            if (h[i][i] < lmin) { lmin = h[i][i]; cmin = jrow[i]; }
        L.cvec[tid] = cmin;  // component `tid` of the bottom eigenvector of H
        if (tid == 0) {
            L.lmin = lmin;  // (the Ritz pair's own value: what its residual is taken against)
            L.null_is_min = 0;
            if (L.has_null && lmin > 0.0) {  // the exact zeros of the null rows are the bottom of the spectrum
                lmin = 0.0;
                L.null_is_min = 1;
            }
            L.o_lmin = lmin;
        }
    }
    if (ABORT && tid == 0 && kw != 0.0) L.abort = 1;
    if (ABORT) kw = gld<COH_AGENT>(s + SC_KWIN);
    __syncthreads();
    EV_STAMP8(const_cast<double *>(s), k, 4);
    if (ABORT && L.abort) return 0.0;
    // (the verdict on X_(k-1), which the decision will ask for first: its round trip runs under the residual)
    double prev_v = 0.0;
    if (ABORT && tid == 0 && k > RITZ_K0) prev_v = gld<COH_AGENT>(s + SC_VERD + k - 1 - RITZ_K0);
    // the bottom Ritz pair (theta, u = V c) and its residual |A u - theta u| (A V is in the partials of H)
    double u = 0.0;
    if (tid < SN) {
        double au = 0.0;
#pragma unroll
        for (int q = 0; q < RITZ; ++q) {
            u = fma(L.cvec[q], L.V[q][tid], u);
            au = fma(L.cvec[q], (L.AVp[0][q][tid] + L.AVp[1][q][tid]) + (L.AVp[2][q][tid] + L.AVp[3][q][tid]), au);
        }
        const double d = fma(-L.lmin, u, au);
        const double r2 = wr::wave64_allsum(d * d);
        // the dominance test's raw material (below): (X_k u) at the picked coordinates = u^T (column q of X_k), and u there
#pragma unroll
        for (int q = 0; q < RITZ; ++q) {
            const double nq = wr::wave64_allsum(u * L.C[q][tid]);
            if (lane == 0) L.numq[wave][q] = nq;
            if (tid == L.pick[q]) L.uq[q] = u;
        }
        // Gershgorin bound of B = A + delta I from the row sums of A (only the diagonal term changes), for delta0 = 1e-2 - theta:
        // the delta used below is never smaller, and the bound grows by at most the difference
        const double m = wr::wave64_allmax(in.ra - fabs(in.dg) + fabs(in.dg + (1e-2 - L.o_lmin)));
        if (lane == 0) {
            L.r2[wave] = r2;
            L.red[wave] = m;
        }
    }
    __syncthreads();
    // An iterate that is taken although its pair has NOT converged (the filter's last one: the cap of 16 squarings -- bottoms whose
    // gaps are ~1e-8 of the spectrum's width, e.g. covo-offline's table on `hovering`: lambda_max 9e4, gaps 4e-3) over-estimates
    // lambda_min by up to its residual; B = A + delta I would then be INDEFINITE (NaN from its factorisation) as soon as that
    // exceeds 1e-2.  There is an eigenvalue within |residual| of theta: theta - |residual| is used instead -- B stays positive
    // definite, its floor is then 1e-2 .. 1e-2 + |residual| instead of 1e-2.  A converged pair (residual <= 1e-8 gap) is not touched.
    const double resid2 = L.r2[0] + L.r2[1];
    const double gap = NS_DEFL_SAFETY * L.gap, rtol = NS_DEFL_RESID * gap;
    const bool small_resid = !L.null_is_min && gap > NS_DEFL_MIN_GAP && resid2 <= rtol * rtol;
    const double lmin_used = (small_resid || L.null_is_min) ? L.o_lmin : L.o_lmin - resid2 * qm::rsq64_(fmax(resid2, 1e-300));
    const double delta = -lmin_used + 1e-2;  // covo.py:120-122: offset = -min_eign + 1e-2
    if (tid == 0) {
        const double fro2 = fma((double)SN * delta, delta, fma(2.0 * delta, in.tr, in.f2));  // |A + delta I|_F^2
        const double gersh = fmax(L.red[0], L.red[1]) + (delta - (1e-2 - L.o_lmin));
        const double scale = fmin(gersh, fro2 * qm::rsq64_(fro2)) * (1.0 + 1e-12);  // >= lambda_max(B): eigenvalues of Y0 in (0, 1]
        // the bottom eigenpair is CONVERGED when its residual is small against the gap bound: the evaluation then "passes" (the
        // chain stops here) and the iterations may deflate the pair (see the header)
        const bool conv = small_resid && 1e-2 + gap < 0.25 * scale;
        // A small residual says u is AN eigenvector of A, not that it is the bottom one: four eigenvectors concentrated on single
        // coordinates, with eigenvalues just above a delocalised bottom eigenvector's, take all four picks for a squaring or two and
        // are exact eigenpairs.  The bottom eigenvector is the DOMINANT one of X_k (the filter amplifies nothing more than
        // lambda_min): rho = (X_k u)_j / u_j at the picked coordinate j where u is largest must be X_k's top eigenvalue, ~sqrt(|X_k|_F^2)
        // -- every other eigenvector's is at most 0.32 of it while e_k < 0.05.  Only such a pair passes early.
        int qb = 0;
#pragma unroll
        for (int q = 1; q < RITZ; ++q)
            if (fabs(L.uq[q]) > fabs(L.uq[qb])) qb = q;
        const double rho = (L.numq[0][qb] + L.numq[1][qb]) / L.uq[qb];
        const bool dominant = rho > 0.0 && rho * rho >= 0.25 * L.nrm[k];
        // (... and only an iterate whose norm is already in the gap bound's linear range: what lets the scan launch skip the others)
        L.o_pass = (conv && dominant && t_k > NS_SQ_TGUARD && 0.5 * (1.0 - L.nrm[k]) < RITZ_PASS_E) ? 1 : 0;
        L.o_lmin = lmin_used;
        L.o_scale = scale;
        L.o_lo = 1e-2;
        L.o_gam = 0.0;
        L.o_zc = 0.0;
        if (conv) {
            const double lo = 1e-2 + gap;
            const double ls = lo * scale;
            const double tau = ls * qm::rsq64_(ls);  // sqrt(lo scale): anywhere inside [lo, scale] serves
            L.o_lo = lo;
            L.o_gam = (tau - 1e-2) / scale;
            L.o_zc = scale * qm::rsq64_(scale) * (10.0 - qm::rsq64_(tau));  // sqrt(scale) ((1e-2)^(-1/2) - tau^(-1/2))
        }
        L.o_gapest = L.gap;
        L.o_resid2 = resid2;  // squared
        L.prev_hint = (int)prev_v;
    }
    __syncthreads();
    return u;
}
// the evaluation of X_k becomes the chain's result (slots read by the launches that follow; written coherently: the chain's
// workgroups poll SC_SQ_DONE)
__device__ __forceinline__ void ritz_publish(const RitzLds &L, double u, double *s, int k, int deflate)
{
    const int tid = threadIdx.x;
    if (tid < SN) gst<COH_AGENT>(s + SC_U + tid, u);
    if (tid == 0) {
        const bool defl = deflate && L.o_gam != 0.0;
        gst<COH_AGENT>(s + SC_LMIN, L.o_lmin);
        gst<COH_AGENT>(s + SC_DELTA, -L.o_lmin + 1e-2);
        gst<COH_AGENT>(s + SC_SCALE, L.o_scale);
        gst<COH_AGENT>(s + SC_LO, defl ? L.o_lo : 1e-2);
        gst<COH_AGENT>(s + SC_GAM, defl ? L.o_gam : 0.0);
        gst<COH_AGENT>(s + SC_ZCOEF, defl ? L.o_zc : 0.0);
        gst<COH_AGENT>(s + SC_GAPEST, L.o_gapest);
        gst<COH_AGENT>(s + SC_RESID, L.o_resid2);
        gst<COH_AGENT>(s + SC_KWIN, (double)k);
    }
    // every store above acknowledged (write-through), then the flag the merged launch's Newton-Schulz workgroups poll (ns_wait_result)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) gst<COH_AGENT>(s + SC_READY, 1.0);
}
// The verdict on X_k, in k order (kwin = the FIRST k that passes, whatever order the evaluations finish in): waits for the verdict
// on X_(k-1); `final` = X_k is the filter's last iterate (taken if nothing passed before).  Returns true when the chain's result
// is decided (by this k or an earlier one): the caller leaves.  Uniform over the workgroup.
__device__ __forceinline__ bool ritz_decide(RitzLds &L, double u, double *s, int k, bool final, int deflate)
{
    if (threadIdx.x == 0) {
        int prev = 1;
        if (k > RITZ_K0 && L.prev_hint != 0) prev = L.prev_hint;
        else if (k > RITZ_K0) {
            const long long t0 = wall_clock64();
            while ((prev = (int)gld<COH_AGENT>(s + SC_VERD + k - 1 - RITZ_K0)) == 0) {
                // (an evaluation that notices a taken result at one of its seams leaves WITHOUT a verdict: the taken result
                // itself is the other thing to look for)
                if (gld<COH_AGENT>(s + SC_KWIN) != 0.0) {
                    prev = 2;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 20000000LL) {
                    gst<COH_AGENT>(s + SC_BARFAIL, 1.0);
                    prev = 2;  // leave; the finalize launch turns the flag into NaN outputs
                    break;
                }
            }
        }
        const int d = (prev == 2) ? 2 : ((L.o_pass || final) ? 1 : 0);  // 2: decided before; 1: this one is taken; 0: not taken
        L.decide = d;
        // the flags first (nothing in this launch reads the results themselves; the launches that do come after its end): the
        // chain stops (it may be two squarings further by now), the evaluation of X_(k+1) has its answer
        if (d == 1) gst<COH_AGENT>(s + SC_SQ_DONE, 1.0);
        gst<COH_AGENT>(s + SC_VERD + k - RITZ_K0, d == 0 ? 1.0 : 2.0);
    }
    __syncthreads();
    const int d = L.decide;
    if (d == 1) ritz_publish(L, u, s, k, deflate);
    return d != 0;
}

// every k of every matrix in ONE launch (the paths whose squarings are launches of their own / run batched): workgroup
// (k - RITZ_K0) + RITZ_NK * b -- X_(k-1)'s workgroup has the id before it, so the verdict it waits for is always being worked on
// (dispatch is in id order; the wait is bounded anyway)
__global__ __launch_bounds__(256) void ns_ritz_scan_kernel(const double *__restrict__ Aall, const XBufs xb, double *__restrict__ scall,
                                                           const int deflate, const int final_only)
{
    __shared__ RitzLds L;
    const int b = blockIdx.x / RITZ_NK, k = RITZ_K0 + (int)(blockIdx.x % RITZ_NK), tid = threadIdx.x;
    double *s = scall + (size_t)b * SC_COUNT;
    const int k_final = (int)s[SC_SQ];  // (a launch boundary ago)
    if (k > k_final) return;
    const bool final = k == k_final;
    if (final_only) {  // timing reference (COVO_NS_RITZ_INSIDE=2): rounds 1-4's rule -- the filter's last iterate, nothing else
        if (tid == 0 && k == k_final - 1) gst<COH_AGENT>(s + SC_VERD + k - RITZ_K0, 1.0);
        if (!final) return;
    }
    // a cheap "not taken": the bounded part of the spectrum is still there, or the norm is not yet in the range in which an
    // evaluation can pass (ritz_eval: o_pass) -- an iterate that cannot pass and is not the last
    const double e_k = 0.5 * (1.0 - slot_sum(s + SC_SQN + k * 64, NS_TILES, tid & 63));
    if (!final && (!(s[SC_SQN + k * 64 + 63] > NS_SQ_TGUARD) || !(e_k < RITZ_PASS_E))) {
        if (tid == 0) gst<COH_AGENT>(s + SC_VERD + k - RITZ_K0, 1.0);
        return;
    }
    ritz_picks<COH_NONE>(ns_xk(xb, k - 1) + (size_t)b * SN * SN, s, L);  // (first: its loads are waited for in issue order)
    RitzIn in;
    ritz_load_matrix(in, Aall + (size_t)b * SN * SN, s, tid);
    ritz_load_inputs<COH_NONE>(in, s, tid);
    const double u = ritz_eval<COH_NONE>(in, ns_xk(xb, k) + (size_t)b * SN * SN, s, k, L);
    (void)ritz_decide(L, u, s, k, final, deflate);
}

// Chen-Chow scaled Newton-Schulz: x -> x (a + b x^2) on [l, 1] with a = 1.5 rho, b = -0.5 rho^3, rho^2 = 3/(1 + l + l^2) (equal
// values at both ends of the interval); lambda_min(Y0) = 1e-2/scale, so l_0 = sqrt(1e-2/scale) and l_(k+1) = l_k (a_k + b_k l_k^2)
__device__ __forceinline__ void ns_coef(double l, double &a, double &bq)
{
    const double rho = (l < 1.0 - 1e-9) ? 1.7320508075688772 * qm::rsq64_(1.0 + l + l * l) : 1.0;
    a = 1.5 * rho;
    bq = -0.5 * rho * rho * rho;
}

// ---- Iteration 0 (Z0 = I, Y0 = B/s + gam u u^T):  Y1 = a0 Y0 + b0 Y0^2,  Z1 = a0 I + b0 Y0.  Rounds 1-4 ran it as a launch / phase of its
// own (one product: Y0^2).  Round 5: Y0^2 is ALREADY THERE -- the filter's first squaring formed X_1 = (alpha I - beta A)^2 / n0 -
// I / t_1, i.e. A^2 = (n0 (X_1 + I / t_1) - alpha^2 I + 2 alpha beta A) / beta^2 -- so Y1 and Z1 are affine in {A, X_1, I, u u^T}:
//     Y1 = ya A + yx X_1 + yi I + yu u u^T,      Z1 = za A + zi I + zu u u^T
// (B u = 1e-2 u for the deflated pair): they are written out element-wise, no product (X_1 keeps a buffer of its own for that:
// XBufs::xq) -- by ns_first_elem_kernel, or by the first phase of the one-matrix persistent launch.  Forming them ON LOAD inside
// iteration 1's two products was measured too: + 10 us over two regular batched launches, + 3.3 us on the persistent launch.
struct NsFirst {
    const double *X1, *u;
    double ya, yx, yi, yu, za, zi, zu;
    double a1, b1;  // iteration 1's own coefficients (the table's second entry: the same recurrence, so that nobody waits for the table)
};
__device__ __forceinline__ NsFirst ns_first_setup(const double *X1, const double *s)
{
#pragma clang fp contract(off)  // (every launch that forms these operands must form the same bits)
    NsFirst f;
    const double scale = s[SC_SCALE], delta = s[SC_DELTA], lo = s[SC_LO], gam = s[SC_GAM];
    const double alpha = s[SC_ALPHA], beta = s[SC_BETA], n0 = s[SC_N0], t1 = s[SC_SQN + 64 + 63];
    const double inv = 1.0 / scale, ib2 = 1.0 / (beta * beta), is2 = inv * inv;
    double l = sqrt(lo / scale), a0, b0;
    ns_coef(l, a0, b0);
    l = fmin(1.0, l * fma(b0 * l, l, a0));
    ns_coef(l, f.a1, f.b1);
    // Y0^2 = qx X_1 + qa A + qi I + qu u u^T
    const double qx = n0 * ib2 * is2, qa = (2.0 * alpha * beta * ib2 + 2.0 * delta) * is2;
    const double qi = ((n0 / t1 - alpha * alpha) * ib2 + delta * delta) * is2, qu = gam * (2.0 * (1e-2 * inv) + gam);
    f.ya = fma(b0, qa, a0 * inv);
    f.yx = b0 * qx;
    f.yi = fma(b0, qi, a0 * delta * inv);
    f.yu = fma(b0, qu, a0 * gam);
    f.za = b0 * inv;
    f.zi = fma(b0, delta * inv, a0);
    f.zu = b0 * gam;
    f.X1 = X1;
    f.u = s + SC_U;
    return f;
}
struct LoadY1 {  // applied to an element of A
    NsFirst f;
    __device__ __forceinline__ double operator()(double v, int r, int c) const
    {
#pragma clang fp contract(off)
        return fma(f.ya, v, fma(f.yx, f.X1[(size_t)r * SN + c], fma(f.yu * f.u[r], f.u[c], (r == c) ? f.yi : 0.0)));
    }
};
struct LoadZ1 {  // applied to an element of A
    NsFirst f;
    __device__ __forceinline__ double operator()(double v, int r, int c) const
    {
#pragma clang fp contract(off)
        return fma(f.za, v, fma(f.zu * f.u[r], f.u[c], (r == c) ? f.zi : 0.0));
    }
};
// the coefficient table of the iterations (a serial recurrence, ~1.5 us on one lane: an extra workgroup of iteration 1's first launch)
template <int COH>
__device__ __forceinline__ void ns_coef_table(double *s)
{
    const double scale = s[SC_SCALE], lo = s[SC_LO];
    double l = sqrt(lo / scale);
    for (int k = 0; k < NS_ITERS; ++k) {
        double a, bq;
        ns_coef(l, a, bq);
        gst<COH>(s + SC_COEF + 2 * k, a);
        gst<COH>(s + SC_COEF + 2 * k + 1, bq);
        l = fmin(1.0, l * fma(bq * l, l, a));
    }
}

template <int COH>
__device__ __forceinline__ bool ns_converged(double *s, int iter, int lane, bool writer)
{
    // both loads are issued before either is looked at
    const double done = gld<COH>(s + SC_NS_DONE);
    const double e = (iter >= 2) ? gld<COH>(s + SC_ERR + (iter - 1) * 64 + lane) : 0.0;
    if (done != 0.0) return true;
    if (iter >= 2 && wr::wave64_allsum(e) < NS_TOL2) {
        if (writer) gst<COH>(s + SC_NS_DONE, 1.0);
        return true;
    }
    return false;
}

// ---- Newton-Schulz step k >= 1, part 1:  T = a_k I + b_k Z.Y  (64 tiles; T and T^T are stored).
// Returns false when the iteration has converged (nothing was written).  Workgroup w of matrix b.
template <int COH>
__device__ __forceinline__ bool ns_T_body(const double *Yall, const double *Ztall, double *Tall, double *Ttall, double *scall,
                                          int iter, int b, int w, double (*red)[4][64], double *part)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double *s = scall + (size_t)b * SC_COUNT;
    const size_t off = (size_t)b * SN * SN;
    const int ti = w >> 3, tj = w & 7;
    if (!COH && gridDim.y > 1 && s[SC_NS_DONE] != 0.0) return false;  // batched: the flag before the operands (see ns_square_body)
    TileOps ops;
    tile_load<COH>(ops, Ztall + off, Yall + off, ti, tj, lane, wv, LoadPlain{});  // (Z^T)^T . Y = Z.Y
    const double a = s[SC_COEF + 2 * iter], bq = s[SC_COEF + 2 * iter + 1];  // (the table: ns_first_elem_kernel's extra workgroup)
    if (ns_converged<COH>(s, iter, lane, w == 0 && tid == 0)) return false;  // Y, Z are final
    const f64x4 acc = tile_mma(ops);
    const double p = tile_reduce(acc, red, wv, lane);
    const int row = 16 * ti + (lane >> 4) + 4 * wv, col = 16 * tj + (lane & 15);
    store_both<COH>(Tall + off, Ttall + off, row, col, fma(bq, p, (row == col) ? a : 0.0));
    const double d = p - ((row == col) ? 1.0 : 0.0);
    const double tot = wg_sum4(d * d, part, wv, lane);
    if (tid == 0) gst<COH>(s + SC_ERR + iter * 64 + w, tot);  // |Z Y - I|_F^2 partial
    return true;
}

// ---- part 2:  Y' = Y.T (tiles 0..63),  Z' = T.Z (tiles 64..127); each with its transpose.  Workgroup wx in 0..127.
template <int COH>
__device__ __forceinline__ bool ns_YZ_body(const double *Ytall, const double *Zall, const double *Tall, const double *Ttall,
                                           double *Yout, double *Ytout, double *Zout, double *Ztout, double *scall, int iter,
                                           int zbuf_out, int b, int wx, double (*red)[4][64])
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double *s = scall + (size_t)b * SC_COUNT;
    const bool isZ = wx >= 64;
    const int w = wx & 63;
    const size_t off = (size_t)b * SN * SN;
    const int ti = w >> 3, tj = w & 7;
    // Y' = Y.T : left factor Y -> pass Y^T;   Z' = T.Z : left factor T -> pass T^T
    if (!COH && gridDim.y > 1 && s[SC_NS_DONE] != 0.0) return false;  // batched: the flag before the operands (see ns_square_body)
    TileOps ops;
    tile_load<COH>(ops, (isZ ? Ttall : Ytall) + off, (isZ ? Zall : Tall) + off, ti, tj, lane, wv, LoadPlain{});
    if (ns_converged<COH>(s, iter, lane, false)) return false;  // part 1 of this iteration raised the flag
    if (wx == 0 && tid == 0) {
        gst<COH>(s + SC_ZBUF, (double)zbuf_out);  // which Z buffer holds the newest iterate
        gst<COH>(s + SC_ITERS, (double)(iter + 1));
    }
    const f64x4 acc = tile_mma(ops);
    const double v = tile_reduce(acc, red, wv, lane);
    const int row = 16 * ti + (lane >> 4) + 4 * wv, col = 16 * tj + (lane & 15);
    store_both<COH>((isZ ? Zout : Yout) + off, (isZ ? Ztout : Ytout) + off, row, col, v);
    return true;
}

__global__ __launch_bounds__(256) void ns_T_kernel(const double *__restrict__ Yall, const double *__restrict__ Ztall,
                                                   double *__restrict__ Tall, double *__restrict__ Ttall,
                                                   double *__restrict__ scall, int iter, int batch)
{
    __shared__ double red[4][4][64];
    __shared__ double part[4];
    int b, w;
    if (!ns_block(batch, b, w)) return;
    (void)ns_T_body<COH_NONE>(Yall, Ztall, Tall, Ttall, scall, iter, b, w, red, part);
}

__global__ __launch_bounds__(256) void ns_YZ_kernel(const double *__restrict__ Ytall, const double *__restrict__ Zall,
                                                    const double *__restrict__ Tall, const double *__restrict__ Ttall,
                                                    double *__restrict__ Yout, double *__restrict__ Ytout,
                                                    double *__restrict__ Zout, double *__restrict__ Ztout,
                                                    double *__restrict__ scall, int iter, int zbuf_out, int batch)
{
    __shared__ double red[4][4][64];
    int b, w;
    if (!ns_block(batch, b, w)) return;
    (void)ns_YZ_body<COH_NONE>(Ytall, Zall, Tall, Ttall, Yout, Ytout, Zout, Ztout, scall, iter, zbuf_out, b, w, red);
}

// ---- "iteration 0" of the launch-per-phase and the batched paths: Y1, Z1 (and their transposes) written out element-wise (NsFirst:
// no product) -- grid (64 tiles + one workgroup for the coefficient table, batch); the one-matrix persistent launch does the same as
// its first phase.  Same functors, same bits.
__global__ __launch_bounds__(256) void ns_first_elem_kernel(const double *__restrict__ Aall, const double *__restrict__ X1all,
                                                            double *__restrict__ Yout, double *__restrict__ Ytout,
                                                            double *__restrict__ Zout, double *__restrict__ Ztout,
                                                            double *__restrict__ scall, int zbuf_out, int batch)
{
    int b, w;
    if (!ns_block(batch, b, w)) return;
    const int tid = threadIdx.x;
    double *s = scall + (size_t)b * SC_COUNT;
    if (w == 64) {
        if (tid == 0) ns_coef_table<COH_NONE>(s);
        return;
    }
    const size_t off = (size_t)b * SN * SN;
    const NsFirst f = ns_first_setup(X1all + off, s);
    if (w == 0 && tid == 0) {
        s[SC_ZBUF] = (double)zbuf_out;
        s[SC_ITERS] = 1.0;
    }
    const int ti = w >> 3, tj = w & 7;
    const int row = 16 * ti + (tid >> 4), col = 16 * tj + (tid & 15);
    const double av = Aall[off + (size_t)row * SN + col];
    store_both(Yout + off, Ytout + off, row, col, LoadY1{f}(av, row, col));
    store_both(Zout + off, Ztout + off, row, col, LoadZ1{f}(av, row, col));
}

// ---- batched launches of the Newton-Schulz phases: one workgroup = a 2 x 2 block of 16 x 16 tiles (round 4).  With one tile per
// workgroup a batched phase moves 32 KB of operands per tile through the CU's L1 port (Y.T and T.Z at 32 matrices: 4 096
// workgroups, 134 MB = as many port cycles as the fp64 MFMAs take) -- a 2 x 2 block shares its operands (64 KB for four tiles) and
// needs a quarter of the workgroups.  EVERY tile is computed exactly as the one-tile bodies compute it -- wave q takes the
// K-quarter [32 q, 32 q + 32), 8 MFMAs on an accumulator of its own, the four partial tiles summed (0 + 1) + (2 + 3), the
// per-tile slots filled by the same reductions -- so a matrix of a batch still equals the same matrix alone bit for bit.
struct QuadOps {
    double a[2][8], b[2][8];
};
template <int COH = COH_NONE, class F>
__device__ __forceinline__ void quad_load(QuadOps &o, const double *A, const double *B, int mi, int mj, int lane, int kq, F f)
{
    const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int k = 32 * kq + 4 * kk + hi;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            o.a[h][kk] = f(gld<COH>(A + (size_t)k * SN + 32 * mi + 16 * h + lo), k, 32 * mi + 16 * h + lo);
            o.b[h][kk] = f(gld<COH>(B + (size_t)k * SN + 32 * mj + 16 * h + lo), k, 32 * mj + 16 * h + lo);
        }
    }
}
// the four tiles (ia, ib): this wave's element (row (lane >> 4) + 4 wv, col lane & 15) of each, v[2 ia + ib]
__device__ __forceinline__ void quad_mma_reduce(const QuadOps &o, double (*redq)[4][4][64], int wv, int lane, double v[4])
{
#pragma unroll
    for (int ia = 0; ia < 2; ++ia)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[ia][kk], o.b[ib][kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) redq[2 * ia + ib][wv][r][lane] = acc[r];
        }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t)
        v[t] = (redq[t][0][wv][lane] + redq[t][1][wv][lane]) + (redq[t][2][wv][lane] + redq[t][3][wv][lane]);
}

// part 1 (ns_T_body): grid (16 blocks, batch)
__global__ __launch_bounds__(256) void ns_T_quad_kernel(const double *__restrict__ Yall, const double *__restrict__ Ztall,
                                                        double *__restrict__ Tall, double *__restrict__ Ttall,
                                                        double *__restrict__ scall, int iter, int batch)
{
    __shared__ double redq[4][4][4][64];
    __shared__ double partq[4][4];
    int b, w;
    if (!ns_block(batch, b, w)) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double *s = scall + (size_t)b * SC_COUNT;
    const size_t off = (size_t)b * SN * SN;
    const int mi = w >> 2, mj = w & 3;
    if (s[SC_NS_DONE] != 0.0) return;  // the flag before the operands (see ns_square_body)
    QuadOps ops;
    quad_load(ops, Ztall + off, Yall + off, mi, mj, lane, wv, LoadPlain{});  // (Z^T)^T . Y = Z.Y
    const double a = s[SC_COEF + 2 * iter], bq = s[SC_COEF + 2 * iter + 1];
    if (ns_converged<COH_NONE>(s, iter, lane, w == 0 && tid == 0)) return;
    double p[4];
    quad_mma_reduce(ops, redq, wv, lane, p);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ti = 2 * mi + (t >> 1), tj = 2 * mj + (t & 1);
        const int row = 16 * ti + (lane >> 4) + 4 * wv, col = 16 * tj + (lane & 15);
        store_both(Tall + off, Ttall + off, row, col, fma(bq, p[t], (row == col) ? a : 0.0));
        const double d = p[t] - ((row == col) ? 1.0 : 0.0);
        const double ws = wr::wave64_allsum(d * d);
        if (lane == 0) partq[t][wv] = ws;
    }
    __syncthreads();
    if (tid < 4) {
        const int ti = 2 * mi + (tid >> 1), tj = 2 * mj + (tid & 1);
        s[SC_ERR + iter * 64 + ti * 8 + tj] = (partq[tid][0] + partq[tid][1]) + (partq[tid][2] + partq[tid][3]);
    }
}

// part 2 (ns_YZ_body): grid (32 blocks, batch): blocks 0..15 of Y' = Y.T, 16..31 of Z' = T.Z
__global__ __launch_bounds__(256) void ns_YZ_quad_kernel(const double *__restrict__ Ytall, const double *__restrict__ Zall,
                                                         const double *__restrict__ Tall, const double *__restrict__ Ttall,
                                                         double *__restrict__ Yout, double *__restrict__ Ytout,
                                                         double *__restrict__ Zout, double *__restrict__ Ztout,
                                                         double *__restrict__ scall, int iter, int zbuf_out, int batch)
{
    __shared__ double redq[4][4][4][64];
    int b, wx;
    if (!ns_block(batch, b, wx)) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double *s = scall + (size_t)b * SC_COUNT;
    const bool isZ = wx >= 16;
    const int w = wx & 15, mi = w >> 2, mj = w & 3;
    const size_t off = (size_t)b * SN * SN;
    if (s[SC_NS_DONE] != 0.0) return;
    QuadOps ops;
    quad_load(ops, (isZ ? Ttall : Ytall) + off, (isZ ? Zall : Tall) + off, mi, mj, lane, wv, LoadPlain{});
    if (ns_converged<COH_NONE>(s, iter, lane, false)) return;  // part 1 of this iteration raised the flag
    if (wx == 0 && tid == 0) {
        s[SC_ZBUF] = (double)zbuf_out;
        s[SC_ITERS] = (double)(iter + 1);
    }
    double v[4];
    quad_mma_reduce(ops, redq, wv, lane, v);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ti = 2 * mi + (t >> 1), tj = 2 * mj + (t & 1);
        const int row = 16 * ti + (lane >> 4) + 4 * wv, col = 16 * tj + (lane & 15);
        store_both((isZ ? Zout : Yout) + off, (isZ ? Ztout : Ytout) + off, row, col, v[t]);
    }
}

// ---- the chain's dependent phases inside ONE persistent launch (batch 1 only: the launch's workgroups must be
// co-resident).  A captured graph cannot branch, so as separate launches every phase the caps allow costs 1.6 us even
// after convergence, and a live one ~3.2-3.7 us, most of it launch floor.  Here every workgroup keeps its tile, the phases
// are separated by a barrier across the launch, and all workgroups leave together at the first phase that finds the
// iteration converged (they all read the same slots).  What makes that barrier affordable was measured in
// scripts/probe/barrier_probe.hip and xcd_chain_probe.hip:
//  * cached accesses + agent-scope release/acquire fences (every fence writes back / invalidates the XCD's whole L2): 6-9 us per
//    phase, 15 with 128 workgroups polling -- loses against a launch;
//  * every inter-phase access an agent-scope relaxed atomic (sc1, coherent across the XCDs' L2s; COH_AGENT in the bodies above)
//    and one counter: no fence, 2.9 us per phase, GEMM included (rounds 2-3);
//  * round 4: the launch is CONFINED TO ONE XCD -- the grid is 8 x the workgroups needed and only the linear ids = 0 (mod 8) stay:
//    the dispatcher deals ids round-robin over the XCDs.  Workgroups that share an L2 need no write-through: plain stores,
//    sc1 loads (COH_XCD) and, instead of a counter every workgroup adds to, one flag word per workgroup that one wave polls with a
//    single load: 1.85 us per phase.  Placement is a performance assumption, never a correctness one: every workgroup publishes
//    its XCC id with its first (coherent) flag, everybody sees the same 36 / 64 ids after the first barrier, and only if they are
//    all equal do the later phases switch to COH_XCD; otherwise the launch carries on as in round 3 (COH_AGENT throughout).
// The spin is bounded (0.2 s): a barrier that cannot complete leaves the iteration unconverged instead of hanging the GPU.
#ifndef NS_POLL_SLEEP
#define NS_POLL_SLEEP 1
#endif
__device__ __forceinline__ unsigned ns_xcc_id()
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}
// Returns 0: timed out (the fail flag is raised); 1: passed; 2: passed and every workgroup of the launch reported the XCC id
// `xcc`; + 4: workgroup 0 announced with THIS phase that the launch is to stop (the squaring launch with evaluations inside: the
// stop comes from outside the chain, so it must reach every workgroup at the same phase -- workgroup 0 alone looks at the outside
// flag and passes it on in its flag word, which everybody holds when the barrier opens; a workgroup 0 that stops never writes again).
// flags[w] = (XCC id << 24) | (stop << 23) | phase; nw <= 64 workgroups.  COH as the phase before it stored.
constexpr unsigned NS_FLAG_STOP = 0x800000u, NS_FLAG_PHASE = 0x7fffffu;
template <int COH>
__device__ __forceinline__ int ns_flag_barrier(unsigned *flags, unsigned phase, int w, int nw, unsigned xcc, double *fail_flag,
                                               const double *stop_src = nullptr, const double *guard0 = nullptr,
                                               const double *guard1 = nullptr)
{
    __shared__ int res;
    // (workgroup 0 of a launch that can be stopped from outside: the outside flag is asked for here, behind this phase's stores --
    // its round trip runs under the wait for their acknowledgements)
    double stop_v = 0.0;
    if (stop_src != nullptr && threadIdx.x == 0) {
        stop_v = gld<COH_AGENT>(stop_src);
        // ... and so are the evaluations' "I have what I need of the buffer the next squaring writes" flags (normally long up)
        double g0 = guard0 ? gld<COH_AGENT>(guard0) : 1.0, g1 = guard1 ? gld<COH_AGENT>(guard1) : 1.0;
        const long long t0 = wall_clock64();
        while ((g0 == 0.0 || g1 == 0.0) && stop_v == 0.0) {
            __builtin_amdgcn_s_sleep(NS_POLL_SLEEP);
            stop_v = gld<COH_AGENT>(stop_src);
            if (guard0) g0 = gld<COH_AGENT>(guard0);
            if (guard1) g1 = gld<COH_AGENT>(guard1);
            if (wall_clock64() - t0 > 20000000LL) {
                gst<COH_AGENT>(fail_flag, 1.0);
                stop_v = 1.0;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's stores have been acknowledged (by the L2 / by memory)
    __syncthreads();
    NS_STAMP();  // stores acknowledged
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const unsigned word = (xcc << 24) | ((stop_v != 0.0) ? NS_FLAG_STOP : 0u) | phase;
        if (lane == 0) {
            if (COH == COH_XCD) asm volatile("global_store_dword %0, %1, off" ::"v"(flags + w), "v"(word) : "memory");
            else __hip_atomic_store(flags + w, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const long long t0 = wall_clock64();
        int good = 1;
        unsigned v = (xcc << 24) | phase;
        for (;;) {
            if (lane < nw) v = __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__builtin_amdgcn_ballot_w64((v & NS_FLAG_PHASE) < phase) == 0) break;
            __builtin_amdgcn_s_sleep(NS_POLL_SLEEP);
            if (wall_clock64() - t0 > 20000000LL) {
                good = 0;
                if (lane == 0) gst<COH_AGENT>(fail_flag, 1.0);  // never silently: the finalize launch turns this into NaN outputs
                break;
            }
        }
        const bool one_xcd = __builtin_amdgcn_ballot_w64((v >> 24) != xcc) == 0;
        const unsigned v0 = __builtin_amdgcn_readfirstlane(v);  // workgroup 0's word
        const bool stopped = (v0 & NS_FLAG_PHASE) == phase && (v0 & NS_FLAG_STOP) != 0u;
        if (lane == 0) res = good ? ((one_xcd ? 2 : 1) | (stopped ? 4 : 0)) : 0;
    }
    __syncthreads();
    NS_STAMP();  // barrier passed
    return res;
}

// (matrix, workgroup-in-matrix) of a persistent launch: grid = 8 x nw x ceil(batch / 8); linear id -> XCD id & 7, slot id >> 3;
// matrix b = 8 (slot / nw) + XCD takes workgroups w = slot % nw -- batch 1: only XCD 0's workgroups stay.  Batched (round 4): every
// matrix runs its WHOLE tail at its own pace next to the others (4 per XCD at 32 matrices) instead of paying, launch by launch,
// for the slowest matrix of the batch; when the grid exceeds what is resident, workgroups are dispatched in id order, so the
// matrices ahead of a partially resident one are complete or running and always finish: no deadlock.
__device__ __forceinline__ bool ns_tail_block(int nw, int batch, int &b, int &w, unsigned xcd0 = 0u)
{
    // xcd0 (batch 1 only): the workgroups that stay are the linear ids = xcd0 (mod 8) -- the Newton-Schulz workgroups of the merged
    // launch sit on XCD 1, its squaring chain on XCD 0
    const unsigned slot = blockIdx.x >> 3, xcd = (blockIdx.x - xcd0) & 7u;
    b = (int)(slot / (unsigned)nw) * 8 + (int)xcd;
    w = (int)(slot % (unsigned)nw);
    return b < batch;
}
static inline dim3 ns_tail_grid(int nw, int batch) { return dim3(8 * nw * ((batch + 7) / 8)); }

struct NsBufs {
    double *Y[2], *Yt[2], *Z[2], *Zt[2], *T, *Tt;
    const double *A, *X1;  // iteration 1 forms its operands from the chain's input and the filter's first iterate (NsFirst)
};

// ---- the iteration tail on PAIRS of tiles (round 4): workgroup w of 32 forms the tiles (2 p, tj) and (2 p + 1, tj), p = w >> 3,
// tj = w & 7, of T, then of Y' and of Z'.  The two tiles share their right operand: 48 KB instead of 64 per two tiles, and what a
// phase of the confined launch waits for is ONE L2 (profiles/r04_sigma_batch_l2_counters.log: 2 MB of operands for part 1, 4 MB
// for part 2, against 16 channels x 64 B/clk) -- the MFMA time does not move (32 workgroups x one wave per SIMD x 16 / 32 MFMAs =
// 64 x two waves x 8 / 16), and the barrier has half the workgroups.  Per tile the arithmetic of ns_T_body / ns_YZ_body.
struct PairOps {
    double a[2][8], b[8];
};
template <int COH, class F>
__device__ __forceinline__ void pair_load(PairOps &o, const double *A, const double *B, int p, int tj, int lane, int kq, F f)
{
    const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int k = 32 * kq + 4 * kk + hi;
        o.a[0][kk] = f(gld<COH>(A + (size_t)k * SN + 32 * p + lo), k, 32 * p + lo);
        o.a[1][kk] = f(gld<COH>(A + (size_t)k * SN + 32 * p + 16 + lo), k, 32 * p + 16 + lo);
        o.b[kk] = f(gld<COH>(B + (size_t)k * SN + 16 * tj + lo), k, 16 * tj + lo);
    }
}
// (skip0: tile 0 of the pair is not wanted; v[0] is then meaningless)
__device__ __forceinline__ void pair_mma_reduce(const PairOps &o, double (*redp)[4][4][64], int wv, int lane, double v[2],
                                                bool skip0 = false)
{
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h == 0 && skip0) continue;  // (uniform)
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[h][kk], o.b[kk], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) redp[h][wv][r][lane] = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) v[h] = (redp[h][0][wv][lane] + redp[h][1][wv][lane]) + (redp[h][2][wv][lane] + redp[h][3][wv][lane]);
}
// ---- the squaring launch on pairs (round 4): the lower triangle's 36 tiles as 20 workgroups -- for m = 0..3 the pairs
// {(2m, tj), (2m + 1, tj)}, tj = 0..2m, which share their right operand X(:, tj), and the diagonal tile (2m + 1, 2m + 1) alone
// (it runs the pair's code with tile 0 -- an upper tile -- left out).  0.77 MB of operands per squaring instead of 1.15, 20
// workgroups at the barrier instead of 36.  Per tile the arithmetic of ns_square_body.
constexpr int NS_SQ_PAIR_WG = 20;
// EVAL (evaluations ride in the launch): SC_SQ_DONE is raised from OUTSIDE the chain, at any time -- the chain's workgroups do
// not look at it themselves (they would disagree within a phase); workgroup 0 passes it on through the barrier (ns_flag_barrier)
template <bool FIRST, int COH, bool EVAL = false>
__device__ __forceinline__ bool ns_square_pair_body(const double *X, double *O, double *s, int step, int w,
                                                    double (*redp)[4][4][64], double (*partp)[4])
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double nrm, t_in = 1.0, alpha = 0.0, beta = 0.0;
    // w -> (m, local): group m holds 2m + 2 workgroups (offsets 0, 2, 6, 12)
    const int m = (w >= 12) ? 3 : (w >= 6) ? 2 : (w >= 2) ? 1 : 0;
    const int local = w - m * (m + 1);
    const bool single = local == 2 * m + 1;
    const int tj = single ? 2 * m + 1 : local;
    PairOps ops;
    // (the operands of A are asked for BEFORE the statistics that make Y0 = alpha I - beta A out of them: the chain's first round
    // trip -- A comes from memory, the Hessian's launches wrote it from other XCDs -- runs under the statistics' two barriers)
    pair_load<COH>(ops, X, X, m, tj, lane, wv, LoadPlain{});
    if (FIRST) {
        ns_square_first_stats<COH>(s, w, tid, lane, wv, &redp[0][0][0][0], alpha, beta, nrm);
        const LoadAffine f{alpha, beta};
        const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = 32 * wv + 4 * kk + hi;
            ops.a[0][kk] = f(ops.a[0][kk], k, 32 * m + lo);
            ops.a[1][kk] = f(ops.a[1][kk], k, 32 * m + 16 + lo);
            ops.b[kk] = f(ops.b[kk], k, 16 * tj + lo);
        }
    }
    if (!FIRST) {
        const double done = gld<COH>(s + SC_SQ_DONE);
        const double p1 = (lane < NS_TILES) ? gld<COH>(s + SC_SQN + step * 64 + lane) : 0.0;
        const double p0 = (lane < NS_TILES && step >= 2) ? gld<COH>(s + SC_SQN + (step - 1) * 64 + lane) : 0.0;
        t_in = gld<COH>(s + SC_SQN + step * 64 + 63);
        if (!EVAL && done != 0.0) return false;
        nrm = wr::wave64_allsum(p1);
        if (step >= 2 && t_in > NS_SQ_TGUARD) {
            const double prev = wr::wave64_allsum(p0);
            if (fabs(nrm - prev) <= NS_SQ_TOL * nrm) {
                if (w == 0 && tid == 0) {
                    gst<COH>(s + SC_SQ_FINAL, (double)step);  // X_step is the filter's last iterate
                    gst<COH>(s + SC_SQ_DONE, 1.0);
                }
                return false;
            }
        }
    }
    const double t_out = 2.0 * t_in * t_in * nrm;  // overflows to +inf once the filter has separated: 1 / t_out = 0
    const double inv_t = 1.0 / t_out;
    if (w == 0 && tid == 0) {
        gst<COH>(s + SC_SQ, (double)(step + 1));
        gst<COH>(s + SC_SQN + (step + 1) * 64 + 63, t_out);
    }
    double pv[2];
    pair_mma_reduce(ops, redp, wv, lane, pv, single);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h == 0 && single) continue;  // (uniform)
        const int ti = 2 * m + h;
        const int row = 16 * ti + (lane >> 4) + 4 * wv, col = 16 * tj + (lane & 15);
        const double v = pv[h] * (1.0 / nrm) - ((row == col) ? inv_t : 0.0);
        store_sym<COH>(O, row, col, v);
        const double ws = wr::wave64_allsum((row > col) ? 2.0 * v * v : ((row == col) ? v * v : 0.0));
        if (lane == 0) partp[h][wv] = ws;
    }
    __syncthreads();
    if (tid < 2 && !(tid == 0 && single)) {
        const int ti = 2 * m + tid;
        gst<COH>(s + SC_SQN + (step + 1) * 64 + ti * (ti + 1) / 2 + tj, (partp[tid][0] + partp[tid][1]) + (partp[tid][2] + partp[tid][3]));
    }
    return true;
}
constexpr int NS_SQ_EVAL_WG = 8;        // evaluating workgroups of the one-matrix launch: an evaluation takes three squarings
constexpr int NS_SQ_EVAL_WG_BATCH = 3;  // ... per matrix of a batched launch (its squarings take twice as long; LEAN evaluations)
template <int COH, int NEVAL>
__device__ __forceinline__ void ns_square_tail_pair_rest(const XBufs xb, double *scall, int step_first, int step_last, int w,
                                                         unsigned xcc, double (*redp)[4][4][64], double (*partp)[4])
{
    constexpr bool EVAL = NEVAL > 0;
    constexpr int NW = NS_SQ_PAIR_WG + NEVAL;
    unsigned *flags = reinterpret_cast<unsigned *>(scall + SC_FLAGS);
    unsigned phase = 1;
    for (int step = step_first + 1; step <= step_last; ++step) {  // squaring `step` reads X_step, writes X_(step + 1)
        if (!ns_square_pair_body<false, COH, EVAL>(ns_xk(xb, step), ns_xk(xb, step + 1), scall, step, w, redp, partp)) return;
        // (with evaluations riding along the last iterate, too, is announced by a barrier: the flag words are what they poll)
        if (EVAL || step < step_last) {
            // (the squaring after this barrier, number `phase`, writes X_(phase + 1) over X_(phase - 1): X_(phase - 1)'s evaluation must
            // have its columns and X_phase's evaluation the diagonal -- workgroup 0 holds the barrier for them)
            ++phase;
            const bool gd = EVAL && w == 0;
            const int r = ns_flag_barrier<COH>(flags, phase, w, NW, xcc, scall + SC_BARFAIL, gd ? scall + SC_SQ_DONE : nullptr,
                                               (gd && phase >= 2 && phase <= NS_SQUARINGS) ? scall + SC_HAVE_DIAG + phase - RITZ_K0 : nullptr,
                                               (gd && phase >= 3 && phase <= NS_SQUARINGS + 1) ? scall + SC_HAVE_COLS + phase - 1 - RITZ_K0 : nullptr);
            if (r == 0 || (r & 4)) return;
        }
        if (EVAL && w == 0) EV_STAMP(scall, 48 + step + 1);  // X_(step + 1) complete
    }
    if (EVAL && w == 0 && threadIdx.x == 0) gst<COH_AGENT>(scall + SC_SQ_FINAL, (double)(step_last + 1));  // the cap: X_16 is the last
}
// ---- an evaluating workgroup of the one-matrix squaring launch (e = 0 .. NS_SQ_EVAL_WG - 1): X_k for k = RITZ_K0 + e, + NS_SQ_EVAL_WG, ...
// Polls the chain's barrier flag words (X_k is complete once all of them have reached k): picks its columns on X_(k-1) while X_k is
// being computed, takes them from X_k the moment it is complete, evaluates, decides in k order (ritz_decide) and leaves as soon as
// the chain's result is decided.  The chain works on TWO buffers (XBufs without history): squaring k + 1 overwrites X_k, so workgroup 0
// of the chain holds its barrier until this workgroup has said that it has the diagonal / the columns (SC_HAVE_DIAG / SC_HAVE_COLS;
// normally said microseconds earlier).  Every wait is bounded.
// Waits until the chain's flag words have all reached `phase`.  Returns 1: they have; 0: the evaluation is not wanted any more
// (a result has been taken, the filter stopped before iterate k, a barrier failed) -- leave.  Uniform over the workgroup.
__device__ __forceinline__ int ns_eval_wait(unsigned *flags, double *s, unsigned phase, int k, int *ev_state)
{
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 64) {
        const long long t0 = wall_clock64();
        int st;
        for (;;) {
            const unsigned v = (lane < NS_SQ_PAIR_WG) ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : NS_FLAG_PHASE;
            const double kw = gld<COH_AGENT>(s + SC_KWIN), fin = gld<COH_AGENT>(s + SC_SQ_FINAL), bf = gld<COH_AGENT>(s + SC_BARFAIL);
            const bool ready = __builtin_amdgcn_ballot_w64((v & NS_FLAG_PHASE) < phase) == 0;
            // (the slots are zeroed by the first squaring: trusted only once the chain's own flags are past phase 1)
            const bool live = __builtin_amdgcn_ballot_w64((v & NS_FLAG_PHASE) < 1u) == 0;
            if (live && (kw != 0.0 || bf != 0.0 || (fin != 0.0 && (double)k > fin))) { st = 0; break; }
            if (ready) { st = 1; break; }
            __builtin_amdgcn_s_sleep(NS_POLL_SLEEP);
            if (wall_clock64() - t0 > 20000000LL) {
                if (lane == 0) gst<COH_AGENT>(s + SC_BARFAIL, 1.0);
                st = 0;
                break;
            }
        }
        if (lane == 0) *ev_state = st;
    }
    __syncthreads();
    const int st = *ev_state;
    __syncthreads();
    return st;
}
template <int NEVAL, bool LEAN>
__device__ __forceinline__ void ns_square_evaluator(const double *__restrict__ A, const XBufs xb, double *s, int e, unsigned xcc,
                                                    int deflate, RitzLds &L)
{
    __shared__ int ev_state;
    unsigned *flags = reinterpret_cast<unsigned *>(s + SC_FLAGS);
    const int tid = threadIdx.x;
    // takes part in the first barrier (the placement check reads its XCC id) and never holds up a later one
    if (tid == 0) __hip_atomic_store(flags + NS_SQ_PAIR_WG + e, (xcc << 24) | NS_FLAG_PHASE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    RitzInT<LEAN> in;
    ritz_load_matrix<LEAN>(in, A, s, tid);  // (depends on nothing in this launch: in flight while the first squarings run)
    bool have_in = false;
#pragma unroll 1
    for (int k = RITZ_K0 + e; k <= NS_SQUARINGS; k += NEVAL) {
        // ---- X_(k-1) is complete: the picks
        if (!ns_eval_wait(flags, s, (unsigned)(k - 1), k, &ev_state)) return;
        if (!have_in) {  // (after the first barrier: the first squaring has left the row sums and the bounds)
            ritz_load_inputs<COH_AGENT, LEAN>(in, s, tid);
            have_in = true;
        }
        ritz_picks<COH_AGENT>(ns_xk(xb, k - 1), s, L);
        if (tid == 0) gst<COH_AGENT>(s + SC_HAVE_DIAG + k - RITZ_K0, 1.0);  // (the picks have consumed the diagonal)
        // ---- X_k is complete
        if (!ns_eval_wait(flags, s, (unsigned)k, k, &ev_state)) return;
        EV_STAMP(s, 2 + 3 * (k - 2));
        // an iterate that cannot pass and cannot be the last (the filter stops by itself only beyond the guard, or at the cap)
        if (k < NS_SQUARINGS && !(gld<COH_AGENT>(s + SC_SQN + k * 64 + 63) > NS_SQ_TGUARD)) {
            if (tid == 0) {
                gst<COH_AGENT>(s + SC_HAVE_COLS + k - RITZ_K0, 1.0);
                gst<COH_AGENT>(s + SC_VERD + k - RITZ_K0, 1.0);
            }
            continue;
        }
        const double u = ritz_eval<COH_AGENT, true, LEAN>(in, ns_xk(xb, k), s, k, L, s + SC_HAVE_COLS + k - RITZ_K0);
        if (L.abort) return;
        EV_STAMP(s, 3 + 3 * (k - 2));
        // is X_k the filter's last iterate?  The chain says so when it looks at X_k's norm: SC_SQ_FINAL = k, or squaring k starts
        // (SC_SQ = k + 1).  Only an evaluation that did not pass needs to know.
        if (tid == 0) {
            int st = 1;  // 1: not the last; 2: the last; 0: leave
            if (!L.o_pass) {
                const long long t0 = wall_clock64();
                for (;;) {
                    const double fin = gld<COH_AGENT>(s + SC_SQ_FINAL), sq = gld<COH_AGENT>(s + SC_SQ), kw = gld<COH_AGENT>(s + SC_KWIN);
                    if (kw != 0.0) { st = 0; break; }
                    if (fin != 0.0) { st = ((double)k == fin) ? 2 : ((double)k < fin ? 1 : 0); break; }
                    if (sq >= (double)(k + 1)) { st = 1; break; }
                    __builtin_amdgcn_s_sleep(NS_POLL_SLEEP);
                    if (wall_clock64() - t0 > 20000000LL) {
                        gst<COH_AGENT>(s + SC_BARFAIL, 1.0);
                        st = 0;
                        break;
                    }
                }
            }
            ev_state = st;
        }
        __syncthreads();
        const int st = ev_state;
        if (st == 0) return;
        const bool decided = ritz_decide(L, u, s, k, st == 2, deflate);
        EV_STAMP(s, 4 + 3 * (k - 2));
        if (decided) return;
    }
}
// NEVAL: evaluating workgroups per matrix (0: none -- the evaluations are a launch of their own, ns_ritz_scan_kernel)
// DYN (the merged launch): the reduction buffer and the evaluations' RitzLds live in the launch's dynamic LDS (`dyn`) -- the merged
// kernel's static LDS + the 129 KiB its log-det workgroup needs must stay under 160 KiB
constexpr size_t NS_SQ_DYN_DOUBLES = 2 * 4 * 4 * 64 + 8 + (sizeof(RitzLds) + 7) / 8;
template <int NEVAL, bool LEAN, bool DYN = false>
__device__ __forceinline__ void ns_square_tail_pair_impl(const double *A, const XBufs xb_all, double *scall, int step_first, int step_last,
                                                         int batch, int force_agent, int deflate, double *dyn = nullptr)
{
    double (*redp)[4][4][64];
    double (*partp)[4];
    if constexpr (DYN) {
        redp = reinterpret_cast<double (*)[4][4][64]>(dyn);
        partp = reinterpret_cast<double (*)[4]>(dyn + 2 * 4 * 4 * 64);
    } else {
        __shared__ double redp_s[2][4][4][64];
        __shared__ double partp_s[2][4];
        redp = redp_s;
        partp = partp_s;
    }
    constexpr bool EVAL = NEVAL > 0;
    constexpr int NW = NS_SQ_PAIR_WG + NEVAL;
    int b, w;
    if (!ns_tail_block(NW, batch, b, w)) return;
    // (EVAL: two buffers -- ns_square_evaluator)
    const XBufs xb{xb_all.x0 + (size_t)b * SN * SN, xb_all.x1 + (size_t)b * SN * SN,
                   EVAL ? nullptr : xb_all.hist + (size_t)b * SN * SN, xb_all.xq + (size_t)b * SN * SN, xb_all.M};
    scall += (size_t)b * SC_COUNT;
    const unsigned xcc = ns_xcc_id();
    if (EVAL && w >= NS_SQ_PAIR_WG) {
        if constexpr (DYN) {
            RitzLds &L = *reinterpret_cast<RitzLds *>(dyn + 2 * 4 * 4 * 64 + 8);
            ns_square_evaluator<NEVAL, LEAN>(A + (size_t)b * SN * SN, xb, scall, w - NS_SQ_PAIR_WG, xcc, deflate, L);
        } else {
            __shared__ RitzLds L;
            ns_square_evaluator<NEVAL, LEAN>(A + (size_t)b * SN * SN, xb, scall, w - NS_SQ_PAIR_WG, xcc, deflate, L);
        }
        return;
    }
    if (EVAL && w == 0) EV_STAMP(scall, 0);
    if (step_first == 0) {
        (void)ns_square_pair_body<true, COH_AGENT>(A + (size_t)b * SN * SN, ns_xk(xb, 1), scall, 0, w, redp, partp);
    } else if (!ns_square_pair_body<false, COH_AGENT>(ns_xk(xb, step_first), ns_xk(xb, step_first + 1), scall, step_first, w, redp, partp))
        return;
    if (step_first == step_last) return;
    int r = ns_flag_barrier<COH_AGENT>(reinterpret_cast<unsigned *>(scall + SC_FLAGS), 1u, w, NW, xcc, scall + SC_BARFAIL);
    if (r == 2 && force_agent) r = 1;
    if (w == 0 && threadIdx.x == 0) scall[SC_PROF + 5] = (double)r;  // diagnostics: which mode the squaring launch ran in
    if (r == 2) ns_square_tail_pair_rest<COH_XCD, NEVAL>(xb, scall, step_first, step_last, w, xcc, redp, partp);
    else if (r == 1) ns_square_tail_pair_rest<COH_AGENT, NEVAL>(xb, scall, step_first, step_last, w, xcc, redp, partp);
    if (EVAL && w == 0) EV_STAMP(scall, 1);
}
template <int NEVAL>
__global__ __launch_bounds__(256) void ns_square_tail_pair_kernel(const double *A, const XBufs xb_all, double *scall, int step_first,
                                                                  int step_last, int batch, int force_agent, int deflate)
{
    ns_square_tail_pair_impl<NEVAL, false>(A, xb_all, scall, step_first, step_last, batch, force_agent, deflate);
}
// batched, evaluations inside: four matrices per XCD = 92 workgroups on its 32 CUs -- three per CU, i.e. three waves per SIMD
// (<= 168 VGPRs: the LEAN evaluation keeps A in memory)
__global__ __launch_bounds__(256, 3) void ns_square_tail_pair_lean_kernel(const double *A, const XBufs xb_all, double *scall, int step_first,
                                                                          int step_last, int batch, int force_agent, int deflate)
{
    ns_square_tail_pair_impl<NS_SQ_EVAL_WG_BATCH, true>(A, xb_all, scall, step_first, step_last, batch, force_agent, deflate);
}

constexpr int NS_PAIR_WG = 32;
template <int COH>
__device__ __forceinline__ bool ns_T_pair_body(const double *Y, const double *Zt, double *T, double *Tt, double *s, int iter, int w,
                                               double (*redp)[4][4][64], double (*partp)[4])
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int p = w >> 3, tj = w & 7;
    PairOps ops;
    pair_load<COH>(ops, Zt, Y, p, tj, lane, wv, LoadPlain{});  // (Z^T)^T . Y = Z.Y
    const double a = gld<COH>(s + SC_COEF + 2 * iter), bq = gld<COH>(s + SC_COEF + 2 * iter + 1);
    if (ns_converged<COH>(s, iter, lane, w == 0 && tid == 0)) return false;
    NS_STAMP();  // operands + slots have arrived
    double pv[2];
    pair_mma_reduce(ops, redp, wv, lane, pv);
    NS_STAMP();  // products reduced
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ti = 2 * p + h;
        const int row = 16 * ti + (lane >> 4) + 4 * wv, col = 16 * tj + (lane & 15);
        store_both<COH>(T, Tt, row, col, fma(bq, pv[h], (row == col) ? a : 0.0));
        const double d = pv[h] - ((row == col) ? 1.0 : 0.0);
        const double ws = wr::wave64_allsum(d * d);
        if (lane == 0) partp[h][wv] = ws;
    }
    __syncthreads();
    if (tid < 2) gst<COH>(s + SC_ERR + iter * 64 + (2 * p + tid) * 8 + tj, (partp[tid][0] + partp[tid][1]) + (partp[tid][2] + partp[tid][3]));
    return true;
}
// Part 2 on 2 x 2 tile BLOCKS (round 5): workgroups 0..15 form the blocks of Y' = Y.T, 16..31 those of Z' = T.Z -- one product of four
// tiles per workgroup instead of two products of two tiles one after the other.  What this phase waits for is the XCD's ONE L2
// (16 channels x 64 B/clk): pairs pull 4 MB of operands through it (~2 us: the stamps of -DNS_STAMPS show the second product's
// operands arriving 2.0 us after the first's), blocks 2 MB; the MFMA count per wave is the same (32).  Per tile the arithmetic of
// ns_YZ_body (quad_mma_reduce), as in the batched launches.  redq: 4 x 4 x 4 x 64 doubles of LDS.  Measured (-DNS_STAMPS, one matrix): the
// phase 3.9 us against 4.2 (operands 1.4 us against 1.2 + the second product's 1.2 under its own product; product 1.3 against 2.0;
// four tiles' stores 0.8 against 0.5); same box, whole step 173.3 against 175.2 us.  (Round 4 had tried all 48 operand loads of the
// two pair products in flight together: no difference -- the bytes, not the second latency.)
template <int COH>
__device__ __forceinline__ bool ns_YZ_quad_body(const double *Yt, const double *Z, const double *T, const double *Tt, double *Yo, double *Yto,
                                                double *Zo, double *Zto, double *s, int iter, int zbuf_out, int w,
                                                double (*redq)[4][4][64])
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool isZ = w >= 16;
    const int q = w & 15, mi = q >> 2, mj = q & 3;
    QuadOps ops;
    quad_load<COH>(ops, isZ ? Tt : Yt, isZ ? Z : T, mi, mj, lane, wv, LoadPlain{});
    if (ns_converged<COH>(s, iter, lane, false)) return false;
    if (w == 0 && tid == 0) {
        gst<COH>(s + SC_ZBUF, (double)zbuf_out);
        gst<COH>(s + SC_ITERS, (double)(iter + 1));
    }
    NS_STAMP();  // operands + slots have arrived
    double v[4];
    quad_mma_reduce(ops, redq, wv, lane, v);
    NS_STAMP();  // products reduced
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ti = 2 * mi + (t >> 1), tj = 2 * mj + (t & 1);
        const int row = 16 * ti + (lane >> 4) + 4 * wv, col = 16 * tj + (lane & 15);
        store_both<COH>(isZ ? Zo : Yo, isZ ? Zto : Yto, row, col, v[t]);
    }
    return true;
}
template <int COH>
__device__ __forceinline__ void ns_iter_tail_pair_rest(const NsBufs &B, size_t off, double *scall, int iter_first, int iter_last, int w,
                                                       unsigned xcc, double (*redp)[4][4][64], double (*partp)[4], bool t_first)
{
    // t_first: part 1 of iteration iter_first is still to do (the launch began by writing Y1, Z1 out); else the launch's first phase was it
    unsigned *flags = reinterpret_cast<unsigned *>(scall + SC_FLAGS) + 64;
    unsigned phase = 1;
    for (int iter = iter_first; iter <= iter_last; ++iter) {
        const bool odd = (iter & 1) != 0;
        const double *Yi = (odd ? B.Y[1] : B.Y[0]) + off, *Yti = (odd ? B.Yt[1] : B.Yt[0]) + off;
        const double *Zi = (odd ? B.Z[1] : B.Z[0]) + off, *Zti = (odd ? B.Zt[1] : B.Zt[0]) + off;
        double *Yo = (odd ? B.Y[0] : B.Y[1]) + off, *Yto = (odd ? B.Yt[0] : B.Yt[1]) + off;
        double *Zo = (odd ? B.Z[0] : B.Z[1]) + off, *Zto = (odd ? B.Zt[0] : B.Zt[1]) + off;
        if (iter > iter_first || t_first) {
            if (!ns_T_pair_body<COH>(Yi, Zti, B.T + off, B.Tt + off, scall, iter, w, redp, partp)) return;
            if (!ns_flag_barrier<COH>(flags, ++phase, w, NS_PAIR_WG, xcc, scall + SC_BARFAIL)) return;
        }
        (void)ns_YZ_quad_body<COH>(Yti, Zi, B.T + off, B.Tt + off, Yo, Yto, Zo, Zto, scall, iter, odd ? 0 : 1, w, redp);
        if (iter < iter_last && !ns_flag_barrier<COH>(flags, ++phase, w, NS_PAIR_WG, xcc, scall + SC_BARFAIL)) return;
    }
}
template <int NWAVES>
__device__ void ns_logdetB_workgroup(const double *__restrict__ A, double *__restrict__ s, double *__restrict__ sm, double *__restrict__ red);
// with_table (one matrix, every iteration folded: iter_first = 1): the coefficient table of the iterations to come -- a serial
// recurrence, ~1.5 us on one lane -- is the work of one of the workgroups the launch would send away anyway (linear id 1: another
// XCD, so it publishes coherently and raises SC_BAR); iteration 1 itself does not need it (NsFirst carries its coefficients)
// early_logdet (one matrix): linear id 2 factors B = A + delta I for its log det (ns_logdetB_workgroup; the launch then carries
// 129 KiB of dynamic LDS: one workgroup per CU, which is how the XCD's 32 CUs host the 32 workgroups of the iterations anyway)
// (a template argument: the factorisation's registers -- 184 against 122 -- must not cost the batched launches their third wave per SIMD)
// MERGED launch (round 6, one matrix): the squaring chain with its evaluations (XCD 0) and the Newton-Schulz workgroups (XCD 1; log det
// on XCD 2, the coefficient table on XCD 3) are ONE launch -- ns_chain_kernel below.  The Newton-Schulz side is resident, placed and
// waiting (ns_wait_result: SC_READY, raised by ritz_publish behind its acknowledged write-through stores) when the evaluation that
// decides arrives: the launch boundary, the dispatch and the launch's ramp are off the chain.  Same arithmetic, same bits.  (The
// same overlap with TWO launches on two streams cost 16 us per step in cross-stream fork + join: scripts/probe/ns_concurrent_spike.patch.)
// Bounded like every wait of the chain (0.2 s -> SC_BARFAIL -> NaN outputs + sticky status).
__device__ __forceinline__ bool ns_wait_result(double *s)
{
    __shared__ int ready_ok;
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        int good = 1;
        while (gld<COH_AGENT>(s + SC_READY) == 0.0) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > 20000000LL) {
                good = 0;
                gst<COH_AGENT>(s + SC_BARFAIL, 1.0);
                break;
            }
        }
        ready_ok = good;
    }
    __syncthreads();
    return ready_ok != 0;
}
template <bool EARLY_LOGDET>
__device__ __forceinline__ void ns_iter_tail_pair_impl(const double *A, const NsBufs B, double *scall, int iter_first, int iter_last,
                                                       int batch, int force_agent, int with_table, int merged, double *ld_sm)
{
    // dynamic LDS: the factorisation's matrix (129 KiB, EARLY_LOGDET) / the iterations' reduction buffer (32 KiB: four tiles x four
    // K-quarters, ns_YZ_quad_body; the pairs of part 1 use half of it) -- static and dynamic together must stay under 160 KiB
    double (*redp)[4][4][64] = reinterpret_cast<double (*)[4][4][64]>(ld_sm);
    __shared__ double partp[2][4];
    if (EARLY_LOGDET && blockIdx.x == 2) {  // (another of the linear ids the launch sends away: XCD 2)
        if (merged && !ns_wait_result(scall)) return;
        ns_logdetB_workgroup<4>(A, scall, ld_sm, &partp[0][0]);
        return;
    }
    if (with_table && blockIdx.x == (merged ? 3u : 1u)) {  // (merged: XCD 1 hosts the iterations)
        if (merged && !ns_wait_result(scall)) return;
        if (threadIdx.x == 0) {
            ns_coef_table<COH_AGENT>(scall);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            gst<COH_AGENT>(scall + SC_BAR, 1.0);
        }
        return;
    }
    int b, w;
    if (!ns_tail_block(NS_PAIR_WG, batch, b, w, merged ? 1u : 0u)) return;
    const size_t off = (size_t)b * SN * SN;
    scall += (size_t)b * SC_COUNT;
    const unsigned xcc = ns_xcc_id();
    const bool odd = (iter_first & 1) != 0;
    if (merged && !ns_wait_result(scall)) return;
#ifdef NS_STAMPS
    if (threadIdx.x == 0) g_nstamp = 0;
    __syncthreads();
    NS_STAMP();
#endif
    // the launch's first phase (agent-scope stores: the placement check comes with the barrier behind it)
    if (with_table) {
        // every iteration folded: Y1 and Z1 written out once (element-wise: NsFirst, no product), so that iteration 1's two phases
        // load plain tiles like every other iteration's.  (The batched and the launch-per-phase paths form the same values on load
        // instead -- for them a phase more is a launch more; here forming them on load cost 3.3 us over the phase it saved.)
        const NsFirst f = ns_first_setup(B.X1 + off, scall);
        const LoadY1 fy{f};
        const LoadZ1 fz{f};
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, p = w >> 3, tj = w & 7;
        if (w == 0 && tid == 0) {
            gst<COH_AGENT>(scall + SC_ZBUF, 1.0);
            gst<COH_AGENT>(scall + SC_ITERS, 1.0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = 16 * (2 * p + h) + (lane >> 4) + 4 * wv, col = 16 * tj + (lane & 15);
            const double av = (B.A + off)[(size_t)row * SN + col];
            store_both<COH_AGENT>(B.Y[1] + off, B.Yt[1] + off, row, col, fy(av, row, col));
            store_both<COH_AGENT>(B.Z[1] + off, B.Zt[1] + off, row, col, fz(av, row, col));
        }
    } else if (!ns_T_pair_body<COH_AGENT>((odd ? B.Y[1] : B.Y[0]) + off, (odd ? B.Zt[1] : B.Zt[0]) + off, B.T + off, B.Tt + off, scall,
                                          iter_first, w, redp, partp))
        return;
    int r = ns_flag_barrier<COH_AGENT>(reinterpret_cast<unsigned *>(scall + SC_FLAGS) + 64, 1u, w, NS_PAIR_WG, xcc, scall + SC_BARFAIL);
    if (r == 2 && force_agent) r = 1;
    if (w == 0 && threadIdx.x == 0) scall[SC_PROF + 6] = (double)r;
    if (with_table) {
        // the coefficient table must be there before part 1 of iteration 2 reads it (it normally is: ~2 us against this launch's ~4)
        __shared__ int tab_ok;
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            int good = 1;
            while (gld<COH_AGENT>(scall + SC_BAR) == 0.0) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 20000000LL) {
                    good = 0;
                    gst<COH_AGENT>(scall + SC_BARFAIL, 1.0);
                    break;
                }
            }
            tab_ok = good;
        }
        __syncthreads();
        if (!tab_ok) return;
    }
    if (r == 2) ns_iter_tail_pair_rest<COH_XCD>(B, off, scall, iter_first, iter_last, w, xcc, redp, partp, with_table != 0);
    else if (r == 1) ns_iter_tail_pair_rest<COH_AGENT>(B, off, scall, iter_first, iter_last, w, xcc, redp, partp, with_table != 0);
#ifdef NS_STAMPS
    if (b == 0 && w == 0 && threadIdx.x == 0)
        for (int i = 0; i < 192; ++i) scall[SC_STAMPS + i] = (i < g_nstamp) ? (double)(g_stamp[i] - g_stamp[0]) : -1.0;
#endif
}
template <bool EARLY_LOGDET>
__global__ __launch_bounds__(256) void ns_iter_tail_pair_kernel(const double *A, const NsBufs B, double *scall, int iter_first, int iter_last,
                                                                int batch, int force_agent, int with_table)
{
    extern __shared__ __attribute__((aligned(16))) double ld_sm[];
    ns_iter_tail_pair_impl<EARLY_LOGDET>(A, B, scall, iter_first, iter_last, batch, force_agent, with_table, 0, ld_sm);
}
// one matrix, every squaring and every iteration folded: linear ids = 0 (mod 8) are the squaring launch's workgroups (chain +
// evaluations), = 1 (mod 8) the iterations', 2 the log-det workgroup, 3 the coefficient table -- 8 x 32 ids, 129 KiB of dynamic LDS
// (one workgroup per CU: the launch wants the GPU to itself like the two it replaces)
template <int NEVAL>
__global__ __launch_bounds__(256) void ns_chain_kernel(const double *A, const XBufs xb_all, const NsBufs B, double *scall, int force_agent,
                                                       int deflate)
{
    extern __shared__ __attribute__((aligned(16))) double ld_sm[];
    static_assert(NS_SQ_DYN_DOUBLES * sizeof(double) <= (size_t)SN * (SN + 1) * sizeof(double), "the squaring side's LDS fits the launch's");
    static_assert(NS_SQ_PAIR_WG + NEVAL <= NS_PAIR_WG, "grid = 8 x NS_PAIR_WG covers the squaring side");
    if ((blockIdx.x & 7u) == 0u) {
        ns_square_tail_pair_impl<NEVAL, false, true>(A, xb_all, scall, 0, NS_SQUARINGS - 1, 1, force_agent, deflate, ld_sm);
        return;
    }
    ns_iter_tail_pair_impl<true>(A, B, scall, 1, NS_ITERS - 1, 1, force_agent, 1, 1, ld_sm);
}

// ---- log det B on its own (round 5).  Sigma = c B^(-1/2) needs log c = 2 log(sigma) + log det B / (2 n) (covo.py:124-128).  Rounds
// 1-4 took log det B from the pivots of chol(Z) -- the LAST thing the chain computes, so nothing scaled by c (the factor of Sigma
// the noise GEMM multiplies with) could leave before the whole factorisation was over.  B = A + delta I is known as soon as the Ritz
// launch has delta, ~60 us earlier: one workgroup factors B itself (chol128_lds_mfma: cond(B) <= 1e6, fp64) next to the
// Newton-Schulz iterations and leaves sum_i log diag(chol B) = log det B / 2 with a flag.  Every path of the chain takes cz from it
// (one definition: a matrix of a batch still equals the same matrix alone bit for bit): the persistent iteration launch of a single
// matrix hosts the workgroup (256 threads: NWAVES = 4; ns_iter_tail_pair_kernel), every other path runs it as a sibling of the
// factoring workgroup inside the finalize launch (512 threads).  sm: [128][129] doubles of LDS; red: >= 4 doubles of LDS.
template <int NWAVES>
__device__ void ns_logdetB_workgroup(const double *__restrict__ A, double *__restrict__ s, double *__restrict__ sm, double *__restrict__ red)
{
    constexpr int THREADS = 64 * NWAVES, LD = SN + 1;
    const int tid = threadIdx.x;
    const double delta = s[SC_DELTA];  // (written by the Ritz launch, a kernel boundary ago)
    const double2 *A2 = reinterpret_cast<const double2 *>(A);
#pragma unroll 4
    for (int e = tid; e < SN * SN / 2; e += THREADS) {  // A is stored exactly symmetric: the full matrix, both triangles
        const int r = e / (SN / 2), c = 2 * (e % (SN / 2));
        const double2 v = A2[e];
        sm[c * LD + r] = v.x + ((r == c) ? delta : 0.0);
        sm[(c + 1) * LD + r] = v.y + ((r == c + 1) ? delta : 0.0);
    }
    __syncthreads();
    chol128_lds_mfma<LD, NWAVES>(sm, tid);
    // sum log diag with ONE libm log: mantissas multiplied (64 factors in [1/2, 1) per wave), exponents added
    if (tid < SN) {
        const double d = sm[tid * LD + tid];
        const double pm = wr::wave64_allprod(__builtin_amdgcn_frexp_mant(d));
        const double pe = wr::wave64_allsum((double)__builtin_amdgcn_frexp_exp(d));
        if ((tid & 63) == 0) {
            red[2 * (tid >> 6)] = pm;
            red[2 * (tid >> 6) + 1] = pe;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const double sumlog = fma(red[1] + red[3], 0.6931471805599453, log(red[0] * red[2]));
        gst<COH_AGENT>(s + SC_SUMLOGB, sumlog);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        gst<COH_AGENT>(s + SC_LDFLAG, 1.0);
    }
}

// ---- one workgroup per matrix: Z ~ sqrt(s) B^(-1/2), symmetrised.  ONE Cholesky serves both needs:
//   Lz = chol(Z)  ->  log det B = n log s - 2 log det Z = n log s - 4 sum log diag(Lz);
//   Sigma = cz Z (cz = c / sqrt(s))  ->  chol(Sigma) = sqrt(cz) Lz.
// Outputs: a_cov = fp32(Sigma) (covo.py:132) and L = fp32(chol(Sigma)).  The reference factors the ROUNDED
// fp32 matrix in fp32 (jax.random.multivariate_normal, covo.py:216); the factor of the unrounded fp64 Sigma
// differs from the exact factor of fp32(Sigma) by <= cond(Sigma) 2^-24 relative, the same size as (and not
// correlated with) the rounding error of the reference's own fp32 factorisation; L L^T reproduces a_cov to
// fp32 rounding (tests/test_gpu_parity.py).
__global__ __launch_bounds__(512) void ns_finalize_kernel(const double *__restrict__ Z0all, const double *__restrict__ Z1all,
                                                          const double *__restrict__ Zt0all, const double *__restrict__ Zt1all,
                                                          double *__restrict__ scall, float sample_sigma,
                                                          float *__restrict__ Sigma_out, float *__restrict__ L_out, int batch,
                                                          const EpsGenArgs gen, int *status, const double *__restrict__ Aall,
                                                          int logdet_here)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double red[512];
    // logdet_here: workgroups 2 b factor B_b = A_b + delta_b I for log det B_b next to the workgroups 2 b + 1 that factor Z_b
    // (ns_logdetB_workgroup); otherwise the Newton-Schulz launch has done it long ago
    // (ids 2 b / 2 b + 1: B_b's workgroup is dispatched BEFORE the one that waits for it -- with a batch beyond what is resident, e.g.
    // covo-offline's 300-row table, workgroups that wait for siblings further down the grid would never let them in)
    if (logdet_here && (int)blockIdx.x < 2 * batch && ((int)blockIdx.x & 1) == 0) {
        const int b = (int)blockIdx.x >> 1;
        ns_logdetB_workgroup<8>(Aall + (size_t)b * SN * SN, scall + (size_t)b * SC_COUNT, sm, red);
        return;
    }
    const int first_passenger = logdet_here ? 2 * batch : batch;
    if ((int)blockIdx.x >= first_passenger) {
        // passenger workgroups (fused step only, eps_tiles.hpp): draw the step's epsilon while workgroups 0..batch-1
        // factor; the dynamic LDS of this launch keeps them at one workgroup (8 waves) per CU
        eps_tiles_generate(gen, ((int)blockIdx.x - first_passenger) * 8 + (int)(threadIdx.x >> 6), ((int)gridDim.x - first_passenger) * 8,
                           (int)(threadIdx.x & 63));
        return;
    }
    const int b = logdet_here ? (int)blockIdx.x >> 1 : (int)blockIdx.x, tid = threadIdx.x;
    double *s = scall + (size_t)b * SC_COUNT;
    // the deflation vector goes to LDS with the launch's FIRST round trip (beside the two scalars the Z loads depend on): read
    // from global memory where it is used, behind `zc != 0`, it was a third dependent round trip of this one-workgroup launch
    __shared__ double suv[SN];
    const double uv_mine = (tid < SN) ? s[SC_U + tid] : 0.0;
    const bool z1 = s[SC_ZBUF] != 0.0;
    const double zc = s[SC_ZCOEF];         // deflation: Z = Z~ + zc u u^T (0: off)
    if (tid < SN) suv[tid] = uv_mine;
    const double *uv = suv;
    const double2 *Z = reinterpret_cast<const double2 *>((z1 ? Z1all : Z0all) + (size_t)b * SN * SN);
    const double2 *Zt = reinterpret_cast<const double2 *>((z1 ? Zt1all : Zt0all) + (size_t)b * SN * SN);
    constexpr int LD = SN + 1;
    const double n = (double)SN;
    long long tk[5];
    tk[0] = clock64();
    // Only the 36 lower 16x16 tiles are fetched (Z and its stored transpose, both coalesced: 8 lanes x 16 B per tile row) and
    // written to both mirror positions -- the factorisation reads the lower triangle and the diagonal blocks as stored.  This
    // one workgroup pulls everything through one CU's L1 (64 B/clk): 144 KiB instead of 256.
    {
        constexpr int TR = NS_TILES * 128 / 512;  // 9 double2 per thread and matrix: all loads in flight first
        double2 za[TR], zt[TR];
#pragma unroll
        for (int t = 0; t < TR; ++t) {
            const int w = __builtin_amdgcn_readfirstlane((tid >> 7) + 4 * t), in = tid & 127;  // tile: uniform over the wave
            int ti, tj;
            tri_tile(w, ti, tj);
            const int e2 = ((16 * ti + (in >> 3)) * SN + 16 * tj + 2 * (in & 7)) >> 1;
            za[t] = Z[e2];
            zt[t] = Zt[e2];
        }
        __syncthreads();  // suv
#pragma unroll
        for (int t = 0; t < TR; ++t) {
            const int w = __builtin_amdgcn_readfirstlane((tid >> 7) + 4 * t), in = tid & 127;
            int ti, tj;
            tri_tile(w, ti, tj);
            const int r = 16 * ti + (in >> 3), c = 16 * tj + 2 * (in & 7);
            double v0 = 0.5 * (za[t].x + zt[t].x), v1 = 0.5 * (za[t].y + zt[t].y);  // covo.py:132 symmetrise
            if (zc != 0.0) {
                const double ur = zc * uv[r];
                v0 = fma(ur, uv[c], v0);
                v1 = fma(ur, uv[c + 1], v1);
            }
            sm[c * LD + r] = v0;        // column-major element (r, c), r >= c up to the diagonal block: what the factorisation reads
            sm[(c + 1) * LD + r] = v1;
            if (ti == tj) {             // diagonal blocks are read as stored: both triangles
                sm[r * LD + c] = v0;
                sm[r * LD + c + 1] = v1;
            }
        }
    }
    __syncthreads();
    tk[1] = clock64();
    chol128_lds_mfma<LD>(sm, tid);
    tk[2] = clock64();
    // covo.py:124-128: log_s = 0.5 log_const - 0.5 log_o, log_const = (2 n 2 log(sigma) + sum log o)/n, i.e. Sigma = c B^(-1/2) with
    // log c = 2 log(sigma) + log det B / (2n); Z = sqrt(scale) B^(-1/2), so Sigma = cz Z with cz = c / sqrt(scale)
    // = sigma^2 exp(sumlogB / n) / sqrt(scale), sumlogB = sum_i log diag(chol B) = log det B / 2 (ns_logdetB_workgroup: a sibling
    // workgroup of this launch, or -- one matrix -- a passenger of the Newton-Schulz launch, long done)
    if (tid == 0) {
        const long long t0 = wall_clock64();
        int good = 1;
        while (gld<COH_AGENT>(s + SC_LDFLAG) == 0.0) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 20000000LL) {  // 0.2 s: the sibling never ran (a starved launch): fail in NaNs, loudly
                good = 0;
                break;
            }
        }
        red[0] = gld<COH_AGENT>(s + SC_SUMLOGB);
        red[1] = good ? 0.0 : 1.0;
    }
    __syncthreads();
    const double sumlogB = red[0];
    const bool ld_failed = red[1] != 0.0;
    // a timed-out grid barrier (ns_grid_barrier) left Z unconverged: fail like the reference's numerical failures do, in NaNs
    const double poison = (s[SC_BARFAIL] != 0.0 || ld_failed) ? __builtin_nan("") : 1.0;
    if (tid == 0 && (s[SC_BARFAIL] != 0.0 || ld_failed) && status != nullptr)  // ... and loudly: the next C call on this handle fails (capi.hip)
        __hip_atomic_fetch_or(status, COVO_DEVSTAT_GRID_BARRIER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const double cz = poison * ((double)sample_sigma * (double)sample_sigma) * exp(sumlogB / n) * qm::rsq64_(s[SC_SCALE]);
    const double sq = cz * qm::rsq64_(cz);
    if (tid == 0) s[SC_CZ] = cz;
    tk[3] = clock64();
    // a_cov = cz sym(Z) needs Z once more: its 32 loads go out first and land while L is written from LDS
    constexpr int TRS = SN * SN / 2 / 512;
    double2 za[TRS], zt[TRS];
    if (Sigma_out) {
#pragma unroll
        for (int t = 0; t < TRS; ++t) {
            za[t] = Z[tid + 512 * t];
            zt[t] = Zt[tid + 512 * t];
        }
    }
    float *Lo = L_out + (size_t)b * SN * SN;
#pragma unroll 4
    for (int e = tid; e < SN * SN; e += 512) {
        const int r = e / SN, c = e % SN;
        Lo[e] = (c <= r) ? (float)(sq * sm[c * LD + r]) : 0.0f;
    }
    if (Sigma_out) {
        float2 *So = reinterpret_cast<float2 *>(Sigma_out + (size_t)b * SN * SN);
#pragma unroll
        for (int t = 0; t < TRS; ++t) {  // a_cov is fp32
            const int e = 2 * (tid + 512 * t), r = e / SN, c = e % SN;
            double v0 = 0.5 * (za[t].x + zt[t].x), v1 = 0.5 * (za[t].y + zt[t].y);
            if (zc != 0.0) {
                const double ur = zc * uv[r];
                v0 = fma(ur, uv[c], v0);
                v1 = fma(ur, uv[c + 1], v1);
            }
            So[tid + 512 * t] = make_float2((float)(cz * v0), (float)(cz * v1));
        }
    }
    tk[4] = clock64();
    if (tid == 0) {
        for (int i = 0; i < 5; ++i) s[SC_PROF + i] = (double)(tk[i] - tk[0]);
    }
}


// ---- the finalize launch with the noise GEMM streamed under the factorisation (round 5; one matrix, fused covo-online step).
// Rounds 2-3 built this twice and reverted it: the GEMM needs the SCALED factor fp32(sqrt(cz) chol(Z)), cz needed log det, log det
// needed every pivot -- so every action had to wait for the end of the factorisation and the launch's 33.5 MB of stores drained
// behind it (54 us fused against 31 + 20 apart).  With log det B known since the Newton-Schulz launch
// (ns_logdetB_workgroup) cz is there before the first pivot:
//   workgroup 0     factors Z exactly as ns_finalize_kernel does; as soon as a 16-column panel is final its waves 1-7 write it,
//                   scaled and rounded to fp32, into L_stream with write-through stores (four complete 64-byte rows per instruction
//                   and wave, issued before the waves turn to the next panel's updates), wait for the acknowledgement where they
//                   would wait for wave 0's F phase anyway, and the last of them raises the panel's flag (the step's sequence
//                   number: no flag is ever cleared);
//   workgroups 1..  the GEMM's workers, noise_gemm_kernel's arithmetic on the same device functions (noise_gemm_body.hpp: same
//                   fragments, same MFMA order -> the same bits): a wave owns one 32-sample tile, draws its epsilon up front (the
//                   launch's first 10 us have nothing else for it), and per 32-column k-group: flag -> the k-group's columns of L
//                   into the LDS image (coherent loads) -> the k-group's MFMAs -> row tile g stored -- final after k-group g, L
//                   being lower triangular -- while workgroup 0 factors on.  The first 32 workers also write a_cov.
// What the launch adds behind the factorisation is one acknowledgement, one flag round trip, one staging and the last panel's
// 8 MFMAs per tile instead of a whole 18 us GEMM launch.
struct StreamHook {
    const double *sm;
    double sq;
    float *Lg;
    unsigned *sync;
    unsigned seq;
    int *cnt;        // LDS [8]: writer waves whose stores of panel p are acknowledged
    double *stamps;
    int tid;
    static constexpr int LD = SN + 1;
    __device__ __forceinline__ void panel_done(int p) const
    {
        const int wave = tid >> 6, lane = tid & 63;
#ifdef NS_STREAM_STAMPS
        if (tid == 0) stamps[2 + p] = (double)wall_clock64();
#endif
        if (wave == 0) return;  // (wave 0 goes straight on to the next diagonal block)
        const int rr = lane >> 4, kk = lane & 15, j0 = 16 * p, c = j0 + kk;
        for (int i0 = j0 + 4 * (wave - 1); i0 < SN; i0 += 28) {
            const int i = i0 + rr;
            const float v = (c <= i) ? (float)(sq * sm[c * LD + i]) : 0.0f;  // ns_finalize_kernel's L_out expression
            __hip_atomic_store(Lg + (size_t)i * SN + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // the stores of panel p - 1 are acknowledged -> its flag (by the last writer wave to get here)
    __device__ __forceinline__ void before_barrier(int p) const
    {
        const int wave = tid >> 6, lane = tid & 63;
        if (p == 0 || wave == 0) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const int old = __hip_atomic_fetch_add(cnt + (p - 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == 6) {
                __hip_atomic_store(sync + p, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sync[1 + (p - 1)]
#ifdef NS_STREAM_STAMPS
                stamps[11 + (p - 1)] = (double)wall_clock64();
#endif
            }
        }
    }
};

// wait until panel P of L_stream carries this step's sequence number (the flags go up in panel order: panel 2 G + 1 = k-group G)
template <int P>
__device__ __forceinline__ bool stream_wait_panel(const StreamGemmArgs &S, unsigned seq, int *flag_lds)
{
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        int good = 1;
        while (__hip_atomic_load(S.sync + 1 + P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 20000000LL) {  // 0.2 s: the factoring workgroup is gone
                good = 0;
                break;
            }
        }
        *flag_lds = good;
    }
    __syncthreads();
    return *flag_lds != 0;
}
// panel P alone: columns [16 P, 16 P + 16), rows from the k-group's first row on (<= 512 float4 chunks: one per thread)
template <int P>
__device__ __forceinline__ void stream_stage_panel(const float *__restrict__ L, float *__restrict__ Ls, int tid)
{
    constexpr int R0 = 32 * (P / 2), CHUNKS = (COVO_NA - R0) * 4;
    if (tid < CHUNKS) {
        const int i = R0 + (tid >> 2), k4 = 16 * P + 4 * (tid & 3);
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(L + (size_t)i * COVO_NA + k4);
        const unsigned long long lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float *d = Ls + i * NG_LDA + k4;
        d[0] = (k4 + 0 <= i) ? __uint_as_float((unsigned)lo) : 0.0f;
        d[1] = (k4 + 1 <= i) ? __uint_as_float((unsigned)(lo >> 32)) : 0.0f;
        d[2] = (k4 + 2 <= i) ? __uint_as_float((unsigned)hi) : 0.0f;
        d[3] = (k4 + 3 <= i) ? __uint_as_float((unsigned)(hi >> 32)) : 0.0f;
    }
}
__global__ __launch_bounds__(512) void ns_finalize_stream_kernel(const double *__restrict__ Z0, const double *__restrict__ Z1,
                                                                 const double *__restrict__ Zt0, const double *__restrict__ Zt1,
                                                                 double *__restrict__ s, float sample_sigma, const StreamGemmArgs S,
                                                                 int *status)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double red[8];
    __shared__ int flag_lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int LD = SN + 1;
    const double n = (double)SN;
    // ---- cz, every workgroup for itself from the same numbers (ns_finalize_kernel's expression: the same bits)
    if (tid == 0) {
        const long long t0 = wall_clock64();
        int good = 1;
        while (gld<COH_AGENT>(s + SC_LDFLAG) == 0.0) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 20000000LL) {
                good = 0;
                break;
            }
        }
        red[0] = gld<COH_AGENT>(s + SC_SUMLOGB);
        red[1] = good ? 0.0 : 1.0;
    }
    __syncthreads();
    const double sumlogB = red[0];
    const bool bad = s[SC_BARFAIL] != 0.0 || red[1] != 0.0;
    const double poison = bad ? __builtin_nan("") : 1.0;
    const double cz = poison * ((double)sample_sigma * (double)sample_sigma) * exp(sumlogB / n) * qm::rsq64_(s[SC_SCALE]);
    const double sq = cz * qm::rsq64_(cz);
    const unsigned seq = __hip_atomic_load(S.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool z1 = s[SC_ZBUF] != 0.0;
    const double zc = s[SC_ZCOEF];  // deflation: Z = Z~ + zc u u^T (0: off)
    // time stamps of the launch (100 MHz wall clock, scripts/stream_timeline.py): workgroup 0 [0] entry, [1] Z in LDS, [2 + p] panel p
    // final, [10] factorisation over, [11 + p] flag p raised; the first and the last worker's wave 0 [24 / 56 + ...]: entry, then per
    // panel {flag seen, staged, multiplied (and stored)}
    // -DNS_STREAM_STAMPS only (make HIPFLAGS+=-DNS_STREAM_STAMPS: scripts/stream_timeline.py); the product kernel stores none
#ifdef NS_STREAM_STAMPS
    double *stamps = s + SC_STAMPS;
#define STREAM_STAMP(i) do { if (tid == 0) stamps[(i)] = (double)wall_clock64(); } while (0)
#else
#define STREAM_STAMP(i) do { } while (0)
#endif

    if (blockIdx.x == 0) {
        // ================================================================ the factoring workgroup
        STREAM_STAMP(0);
        if (tid == 0 && bad && status != nullptr)
            __hip_atomic_fetch_or(status, COVO_DEVSTAT_GRID_BARRIER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (tid == 0) s[SC_CZ] = cz;
        __shared__ double suv[SN];
        __shared__ int wcnt[8];
        if (tid < SN) suv[tid] = s[SC_U + tid];
        if (tid < 8) wcnt[tid] = 0;
        const double2 *Z = reinterpret_cast<const double2 *>(z1 ? Z1 : Z0);
        const double2 *Zt = reinterpret_cast<const double2 *>(z1 ? Zt1 : Zt0);
        long long tk[3];
        tk[0] = clock64();
        {
            constexpr int TR = NS_TILES * 128 / 512;  // 9 double2 per thread and matrix: all loads in flight first
            double2 za[TR], zt[TR];
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                const int w = __builtin_amdgcn_readfirstlane((tid >> 7) + 4 * t), in = tid & 127;
                int ti, tj;
                tri_tile(w, ti, tj);
                const int e2 = ((16 * ti + (in >> 3)) * SN + 16 * tj + 2 * (in & 7)) >> 1;
                za[t] = Z[e2];
                zt[t] = Zt[e2];
            }
            __syncthreads();  // suv
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                const int w = __builtin_amdgcn_readfirstlane((tid >> 7) + 4 * t), in = tid & 127;
                int ti, tj;
                tri_tile(w, ti, tj);
                const int r = 16 * ti + (in >> 3), c = 16 * tj + 2 * (in & 7);
                double v0 = 0.5 * (za[t].x + zt[t].x), v1 = 0.5 * (za[t].y + zt[t].y);  // covo.py:132 symmetrise
                if (zc != 0.0) {
                    const double ur = zc * suv[r];
                    v0 = fma(ur, suv[c], v0);
                    v1 = fma(ur, suv[c + 1], v1);
                }
                sm[c * LD + r] = v0;
                sm[(c + 1) * LD + r] = v1;
                if (ti == tj) {
                    sm[r * LD + c] = v0;
                    sm[r * LD + c + 1] = v1;
                }
            }
        }
        __syncthreads();
        tk[1] = clock64();
        STREAM_STAMP(1);
        StreamHook hook;
        hook.sm = sm;
        hook.sq = sq;
        hook.Lg = S.L_stream;
        hook.sync = S.sync;
        hook.seq = seq;
        hook.cnt = wcnt;
#ifdef NS_STREAM_STAMPS
        hook.stamps = stamps;
#else
        hook.stamps = nullptr;
#endif
        hook.tid = tid;
        chol128_lds_mfma<LD, 8>(sm, tid, hook);
        hook.before_barrier(8);  // the last panel's acknowledgement and flag
        tk[2] = clock64();
        if (tid == 0) {
#ifdef NS_STREAM_STAMPS
            stamps[10] = (double)wall_clock64();
#endif
            for (int i = 0; i < 3; ++i) s[SC_PROF + i] = (double)(tk[i] - tk[0]);
        }
        return;
    }

    // ==================================================================== the GEMM's workers
    // Worker b = blockIdx.x in [1, W] takes the 8 tiles of the stand-alone GEMM's workgroup b -- same XCD (linear id mod 8), so the
    // rollout's XCD-affine mapping finds the stripes in the L2 it expects -- one per wave.  The 8 tiles of "workgroup 0" have no CU
    // (the factoring workgroup sits there): each is hosted by one worker on XCD 0 (b = 8, 16, ...; small launches: b = 1, 2, ...),
    // whose waves draw its epsilon into LDS together and whose waves 1, 2, 3 carry its row tiles {3}, {2}, {0, 1} next to their own
    // tile (64 / 48 / 48 of its 160 MFMAs on accumulators of their own: every dot product still one ascending-k chain).
    float *Ls = reinterpret_cast<float *>(sm);   // [128][NG_LDA]: the LDS image of L, filled k-group by k-group
    float *mus = Ls + COVO_NA * NG_LDA;          // [128]
    float4 *epsx = reinterpret_cast<float4 *>(mus + COVO_NA);  // [16][64]: the hosted tile's epsilon in B-operand order (16 KiB)
    const int b = (int)blockIdx.x, W = (int)gridDim.x - 1;
#ifdef NS_STREAM_STAMPS
    double *wst = (tid == 0 && (b == 1 || b == W)) ? stamps + (b == 1 ? 24 : 56) : nullptr;
#else
    double *const wst = nullptr;
#endif
    if (wst) wst[0] = (double)wall_clock64();
    if (tid < COVO_NA) mus[tid] = S.mu[tid];
    const uint32_t k0 = S.dyn[0], k1 = S.dyn[1];
    const int N = S.N, ntiles = (N + 31) / 32;
    const int nextra = ntiles < 8 ? ntiles : 8;
    const int j = lane & 31, kh = lane >> 5;
    const int t0 = 8 * b + wave;
    const bool has0 = t0 < ntiles;  // (wave-uniform)
    int ex = -1;                    // the tile of "workgroup 0" this worker hosts
    if (W >= 64) { if ((b & 7) == 0 && (b >> 3) - 1 < nextra) ex = (b >> 3) - 1; }
    else if (b - 1 < nextra) ex = b - 1;
    auto id_of = [&](int t) {
        int row = t * 32 + j;
        row = row < N ? row : N - 1;
        return (uint64_t)(S.sample_offset + row);
    };
    const uint64_t id0 = id_of(has0 ? t0 : 0);
    // epsilon of the wave's own tile, all four k-groups, while workgroup 0 loads Z and factors its first panels
    BGroup bq[4];
    if (has0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = gen_group(id0, g, kh, k0, k1);
    }
    if (ex >= 0) {  // the hosted tile's: wave v draws chunks 2 v, 2 v + 1 (eps_tiles.hpp's order: chunk q, lane -> normal4(2 q + kh, sample j))
        const uint64_t idx = id_of(ex);
#pragma unroll
        for (int q = 2 * wave; q < 2 * wave + 2; ++q) epsx[q * 64 + lane] = rngd::normal4((uint32_t)(2 * q + kh), idx, k0, k1);
    }
    const int xrole = ex >= 0 ? wave : 0;  // 1: row tile 3, 2: row tile 2, 3: row tiles 0 and 1 of the hosted tile
    // a_cov = cz sym(Z) (+ the deflated eigenpair): covo.py:132, the expression of ns_finalize_kernel / the noise GEMM's CovDeferred
    if (S.a_cov_out != nullptr && b - 1 < SN * SN / 512) {
        const double *Zb = z1 ? Z1 : Z0, *Ztb = z1 ? Zt1 : Zt0;
        const int wstride = W < SN * SN / 512 ? W : SN * SN / 512;  // (small launches: the workers there are stride over the matrix)
        for (int e = (b - 1) * 512 + tid; e < SN * SN; e += wstride * 512) {
            double v = 0.5 * (Zb[e] + Ztb[e]);
            if (zc != 0.0) v = fma(zc * s[SC_U + e / SN], s[SC_U + e % SN], v);
            S.a_cov_out[e] = (float)(cz * v);
        }
    }
    f32x16 acc0[4], accx[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[rt][e] = 0.0f; accx[rt][e] = 0.0f; }
    const float *__restrict__ La = Ls + j * NG_LDA + kh;
    float4 *__restrict__ a_out = reinterpret_cast<float4 *>(S.a_out);
    bool ok = true;
    auto store_rt = [&](int rt, int tile, const f32x16 (&acc)[4]) {
        const int nn = tile * 32 + j;
        if (nn < N) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int t = 8 * rt + 2 * g + kh;
                const float4 m4 = *reinterpret_cast<const float4 *>(mus + 4 * t);
                float4 v;
                if (S.nanp) {  // COVO_FLAG_PROPAGATE_NAN (covo.py:224 under jnp.clip)
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
                if (!ok) v = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
                // (plain stores: nontemporal / write-through ones take 7.7 us off this launch's end-of-kernel write-back and put
                // 6.1 us on the rollout, which then reads the stripes from HBM instead of its XCD's L2: DESIGN.md 9)
                a_out[(size_t)t * N + nn] = v;
            }
        }
    };
    auto xgroup = [&](int g) {  // the hosted tile's k-group g from LDS, in the lane's B-operand image
        BGroup bx;
#pragma unroll
        for (int i = 0; i < 4; ++i) bx.c[i] = epsx[(4 * g + i) * 64 + lane];
        return bx;
    };
    // panel by panel: a sync point costs ~1 us of flag + staging latency, but a worker that is through with panel p before panel
    // p + 1 is final -- epsilon is in registers, a panel's share of the MFMAs takes less than the panel's factorisation -- pays it
    // while it would wait anyway; behind the LAST panel there are then 8 MFMAs per tile instead of 16
#define STREAM_PANEL(P)                                                                                           \
    do {                                                                                                          \
        constexpr int G_ = (P) / 2, H_ = (P) & 1;                                                                 \
        ok = stream_wait_panel<P>(S, seq, &flag_lds) && ok;                                                       \
        if (wst) wst[1 + 3 * (P)] = (double)wall_clock64();                                                       \
        stream_stage_panel<P>(S.L_stream, Ls, tid);                                                               \
        __syncthreads();                                                                                          \
        if (wst) wst[2 + 3 * (P)] = (double)wall_clock64();                                                       \
        if (has0) mfma_half<G_, H_, 15>(La, bq[G_], acc0);                                                        \
        if (xrole == 1) mfma_half<G_, H_, 8>(La, xgroup(G_), accx);                                               \
        else if (xrole == 2) mfma_half<G_, H_, 4>(La, xgroup(G_), accx);                                          \
        else if (xrole == 3) mfma_half<G_, H_, 3>(La, xgroup(G_), accx);                                          \
        if (H_ == 1) {                                                                                            \
            if (has0) store_rt(G_, t0, acc0);                                                                     \
            if ((G_ == 3 && xrole == 1) || (G_ == 2 && xrole == 2) || (G_ < 2 && xrole == 3)) store_rt(G_, ex, accx); \
        }                                                                                                         \
        if (wst) wst[3 + 3 * (P)] = (double)wall_clock64();                                                       \
    } while (0)
    STREAM_PANEL(0);
    STREAM_PANEL(1);
    STREAM_PANEL(2);
    STREAM_PANEL(3);
    STREAM_PANEL(4);
    STREAM_PANEL(5);
    STREAM_PANEL(6);
    STREAM_PANEL(7);
#undef STREAM_PANEL
    if (!ok && S.a_cov_out != nullptr && b - 1 < SN * SN / 512) {
        // a panel flag timed out: this launch's actions are NaN -- so is the a_cov part this worker wrote before it knew (ADVICE r05:
        // the outputs of a failed step must not disagree; the sticky device status below makes the next call fail)
        const int wstride = W < SN * SN / 512 ? W : SN * SN / 512;
        for (int e = (b - 1) * 512 + tid; e < SN * SN; e += wstride * 512) S.a_cov_out[e] = __builtin_nanf("");
    }
    if (!ok && tid == 0 && status != nullptr)
        __hip_atomic_fetch_or(status, COVO_DEVSTAT_GRID_BARRIER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// how many of the chain's last squarings / Newton-Schulz iterations run inside the persistent launches; covo_debug_set_ns_tail.
// Rounds 2-3 (persistent launches spread over the XCDs, every access sc1, one counter): a folded phase cost nothing once the
// chain had converged (a separate launch: 1.6 us) but, while live, +0.4 us per squaring / +1.4 us per iteration over its
// separate launch(es), so only the often-idle phases were folded (squarings 7.., iterations 6..).
// Round 4 (one XCD, plain stores, flag words; see ns_flag_barrier): a live folded squaring is 0.85 us CHEAPER than its launch, a
// live iteration 0.7 us (scripts/ns_tail_cost.py on a 14-squaring / 9-iteration matrix: covo_sigma 167.5 us with nothing folded,
// 156.5 squarings folded, 161.8 iterations folded, 151.7 both), so batch 1 folds everything but the first squaring and
// iteration 0.  Same box, scripts/tail_bench.py, bench / closed loop steps/s: rounds 2-3 code (9, 6) 5 277-5 305 / 4 685-4 692;
// round 4 code (9, 6) 5 236 / 4 644, (12, 8) 5 280 / 4 738, (15, 8) 5 364-5 384 / 4 811-4 829, (15, 11) 5 370-5 375 / 4 813-4 815.
// Where the rest of a phase goes (workgroup 0's seams, -DNS_STAMPS): part 1 of an iteration = 1.08 us until the operands are in
// (each CU of the one XCD pulls 64 KB through its 64 B/clk port), 0.40 MFMA + reduce, 0.64 until the stores are acknowledged,
// 0.64 flag -> poll; part 2 = 1.36 / 0.84 (2 048 fp64-MFMA cycles per SIMD: a quarter of the phase is now the ONE XCD's matrix
// throughput) / 0.6 / 1.3-1.7 (the other workgroup of the CU finishes its MFMAs later).
// Batched (the env-batched step, covo-offline's table): all squarings are folded -- every matrix runs them at its own pace, 4
// matrices per XCD at 32 -- but only the last 4 iterations: early on every matrix is live, and four matrices' workgroups on an XCD's
// 32 CUs keep its ONE L2 busy 65-80 % of the launch (profiles/r04_sigma_batch_l2_counters.log) where a launch spreads each phase
// over the chip.  bench.py --config envs, control-steps/s / Sigma us per batched step, one tile per workgroup (64 per matrix):
// (0, 0) 71 186 / 288; (15, 0) 73 395 / 274; (15, 3) 74 851 / 266; (15, 5) 75 943 / 255; (15, 7) 73 142 / 274; (15, 11) 72 163 / 282.
// With the 2 x 2 blocks of the batched launches and the pairs of the persistent one (32 workgroups per matrix), bench / closed
// loop: (15, 3) 81 815 / 75 187; (15, 4) 82 553 / 75 641; (15, 5) 82 865 / 74 674; (15, 6) 82 978 / 74 796; (15, 8) 81 889 / 74 194;
// (15, 11) 80 713 / 73 431 -> the last 4.  The squaring launch on pairs too (20 workgroups per matrix instead of 36, a third fewer
// operand bytes): one matrix 5 464-5 470 / 4 874-4 886 against 5 428-5 457 / 4 844-4 854, batched 83 663 / 76 231 against 82 535 / 75 461.
// THE defaults of the four tail lengths (also what covo_debug_set_ns_tail(handle, -1, -1) restores); the other switches of a
// handle (deflation, forced agent-scope coherence, where the Ritz evaluations run): CovoOpts, covo_default_opts (step.hip)
void sigma_ns_tail_defaults(CovoOpts &o)
{
    o.ns_tail_iters = NS_ITERS - 1;
    o.ns_tail_squarings = o.ns_tail_squarings_batched = NS_SQUARINGS - 1;
    o.ns_tail_iters_batched = 4;
}

SymStatsOut sigma_ns_stats_out(void *workspace, int batch)
{
    double *ws = reinterpret_cast<double *>(workspace);
    double *sc = ws + (size_t)11 * batch * SN * SN;
    SymStatsOut o;
    o.rpart = sc + SC_RPART;
    o.fpart = sc + SC_FPART;
    o.diag = sc + SC_DIAG;
    o.stride = SC_COUNT;
    o.flags = sc + SC_FLAGS;
    o.ready = sc + SC_READY;
    return o;
}
// 11 matrices, the slots, then the filter's history X_3 .. X_16 and X_1 (XBufs)
size_t sigma_ns_workspace_bytes(int batch) { return (size_t)batch * ((11 + NS_SQUARINGS - 1) * SN * SN + SC_COUNT) * sizeof(double); }
// (opt.ns_ritz_inside == 0 -- COVO_NS_RITZ_INSIDE=0 / covo_debug_set_ns_ritz_inside(h, 0): the one-matrix chain, too, evaluates after
// its squarings: ns_ritz_scan_kernel)

int launch_sigma_ns(const CovoOpts &opt, const double *R, int batch, float sample_sigma, float *Sigma, float *L, void *workspace,
                    hipStream_t s, const EpsGenArgs *gen, int *status, bool persistent_ok, CovDeferred *cov, bool r_has_stats,
                    const StreamGemmArgs *stream, bool *streamed)
{
    if (streamed != nullptr) *streamed = false;
    double *ws = reinterpret_cast<double *>(workspace);
    const size_t M = (size_t)batch * SN * SN;
    double *A = ws, *X0 = ws + M, *X1 = ws + 2 * M;
    double *Y[2] = {ws + 3 * M, ws + 4 * M}, *Yt[2] = {ws + 5 * M, ws + 6 * M};
    double *Z[2] = {ws + 7 * M, ws + 8 * M}, *Zt[2] = {ws + 9 * M, ws + 10 * M};
    double *T = X0, *Tt = X1;  // the squaring buffers are free once lambda_min is known
    double *sc = ws + 11 * M;
    const size_t lds = (size_t)SN * (SN + 1) * sizeof(double);
    static unsigned long long attr_devices = 0;  // (per device: covo_first_on_device)
    if (covo_first_on_device(attr_devices)) {
        COVO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ns_finalize_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        COVO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ns_iter_tail_pair_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        COVO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ns_chain_kernel<NS_SQ_EVAL_WG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        COVO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ns_finalize_stream_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (r_has_stats) A = const_cast<double *>(R);  // exactly symmetric, statistics already in sc (KD): no prep launch
    else hipLaunchKernelGGL(ns_prep_kernel, ns_grid(NS_TILES, batch), dim3(256), 0, s, R, A, sc, batch);
    double *hist0 = sc + (size_t)batch * SC_COUNT;
    const XBufs xb{X0, X1, hist0, hist0 + (size_t)(NS_SQUARINGS - 2) * M, M};
    const bool fold_first = persistent_ok && (batch == 1 ? opt.ns_tail_squarings : opt.ns_tail_squarings_batched) >= NS_SQUARINGS - 1;
    if (!fold_first) hipLaunchKernelGGL(ns_square_kernel<true>, ns_grid(NS_TILES, batch), dim3(256), 0, s, A, ns_xk(xb, 1), sc, 0, batch);
    // the remaining squarings / iterations run inside persistent launches (20 / 32 workgroups per matrix, one XCD per matrix)
    int sq_tail = persistent_ok ? (batch == 1 ? opt.ns_tail_squarings : opt.ns_tail_squarings_batched) : 0;
    if (sq_tail > NS_SQUARINGS - 1) sq_tail = NS_SQUARINGS - 1;
    const int sq_sep = NS_SQUARINGS - sq_tail;
    for (int i = 1; i < sq_sep; ++i)
        hipLaunchKernelGGL(ns_square_kernel<false>, ns_grid(NS_TILES, batch), dim3(256), 0, s, ns_xk(xb, i), ns_xk(xb, i + 1), sc, i, batch);
    // every squaring folded: the Rayleigh-Ritz evaluations ride in the squaring launch and stop it (ns_square_evaluator)
    const bool eval_inside = fold_first && opt.ns_ritz_inside == 1;
    // round 6: one matrix with both persistent launches whole -> ONE launch (ns_chain_kernel): the iterations' workgroups wait inside
    // for the chain's result.  r_has_stats: SC_READY and the barrier flag words were cleared a launch ahead (KD).
    const bool merged = opt.ns_merged && batch == 1 && persistent_ok && eval_inside && r_has_stats && sq_tail >= NS_SQUARINGS - 1 &&
                        opt.ns_tail_iters >= NS_ITERS - 1 && g_dbg_sigma_stages >= 3;
    if (merged) {
        NsBufs B;
        for (int k = 0; k < 2; ++k) {
            B.Y[k] = Y[k];
            B.Yt[k] = Yt[k];
            B.Z[k] = Z[k];
            B.Zt[k] = Zt[k];
        }
        B.T = T;
        B.Tt = Tt;
        B.A = A;
        B.X1 = ns_xk(xb, 1);
        hipLaunchKernelGGL((ns_chain_kernel<NS_SQ_EVAL_WG>), ns_tail_grid(NS_PAIR_WG, 1), dim3(256), lds, s, A, xb, B, sc, opt.ns_force_agent,
                           opt.ns_deflate);
    } else
    if (sq_tail > 0) {
        if (eval_inside && batch == 1)
            hipLaunchKernelGGL((ns_square_tail_pair_kernel<NS_SQ_EVAL_WG>), ns_tail_grid(NS_SQ_PAIR_WG + NS_SQ_EVAL_WG, batch), dim3(256), 0,
                               s, A, xb, sc, 0, NS_SQUARINGS - 1, batch, opt.ns_force_agent, opt.ns_deflate);
        else if (eval_inside)
            hipLaunchKernelGGL(ns_square_tail_pair_lean_kernel, ns_tail_grid(NS_SQ_PAIR_WG + NS_SQ_EVAL_WG_BATCH, batch),
                               dim3(256), 0, s, A, xb, sc, 0, NS_SQUARINGS - 1, batch, opt.ns_force_agent, opt.ns_deflate);
        else
            hipLaunchKernelGGL((ns_square_tail_pair_kernel<0>), ns_tail_grid(NS_SQ_PAIR_WG, batch), dim3(256), 0, s, A, xb, sc,
                               fold_first ? 0 : sq_sep, NS_SQUARINGS - 1, batch, opt.ns_force_agent, opt.ns_deflate);
    }
    if (g_dbg_sigma_stages < 2) return 0;
    if (!eval_inside && !merged) hipLaunchKernelGGL(ns_ritz_scan_kernel, dim3(batch * RITZ_NK), dim3(256), 0, s, A, xb, sc, opt.ns_deflate, opt.ns_ritz_inside == 2 ? 1 : 0);
    if (g_dbg_sigma_stages < 3) return 0;
    // Iteration 0 is no product (NsFirst): Y1 and Z1 are written out element-wise from A and X_1 -- by the first phase of the persistent
    // launch when one matrix folds every iteration into it, else by a launch of their own (which also makes the coefficient table).
    const double *Xq = ns_xk(xb, 1);  // X_1
    const bool fold_all = batch == 1 && persistent_ok && opt.ns_tail_iters >= NS_ITERS - 1;
    if (!fold_all && !merged) hipLaunchKernelGGL(ns_first_elem_kernel, ns_grid(65, batch), dim3(256), 0, s, A, Xq, Y[1], Yt[1], Z[1], Zt[1], sc, 1, batch);
    bool early_logdet = false;
    int n_tail = persistent_ok ? (batch == 1 ? opt.ns_tail_iters : opt.ns_tail_iters_batched) : 0;
    if (n_tail > NS_ITERS - 1) n_tail = NS_ITERS - 1;
    const int n_sep = NS_ITERS - n_tail;
    for (int i = 1; i < n_sep; ++i) {
        const int in = i & 1, out = in ^ 1;
        if (batch > 1) {  // 2 x 2 tile blocks per workgroup: same tiles, same bits, half the operand traffic
            hipLaunchKernelGGL(ns_T_quad_kernel, ns_grid(16, batch), dim3(256), 0, s, Y[in], Zt[in], T, Tt, sc, i, batch);
            hipLaunchKernelGGL(ns_YZ_quad_kernel, ns_grid(32, batch), dim3(256), 0, s, Yt[in], Z[in], T, Tt, Y[out], Yt[out], Z[out],
                               Zt[out], sc, i, out, batch);
            continue;
        }
        hipLaunchKernelGGL(ns_T_kernel, ns_grid(64, batch), dim3(256), 0, s, Y[in], Zt[in], T, Tt, sc, i, batch);
        hipLaunchKernelGGL(ns_YZ_kernel, ns_grid(128, batch), dim3(256), 0, s, Yt[in], Z[in], T, Tt, Y[out], Yt[out], Z[out],
                           Zt[out], sc, i, out, batch);
    }
    if (n_tail > 0) {
        NsBufs B;
        for (int k = 0; k < 2; ++k) {
            B.Y[k] = Y[k];
            B.Yt[k] = Yt[k];
            B.Z[k] = Z[k];
            B.Zt[k] = Zt[k];
        }
        B.T = T;
        B.Tt = Tt;
        B.A = A;
        B.X1 = Xq;
        // one matrix: log det B rides in this launch (ns_logdetB_workgroup), ~50 us before the finalize launch wants it
        early_logdet = batch == 1;
        if (merged) {
        } else if (early_logdet)
            hipLaunchKernelGGL(ns_iter_tail_pair_kernel<true>, ns_tail_grid(NS_PAIR_WG, batch), dim3(256), lds, s, A, B, sc,
                               fold_all ? 1 : n_sep, NS_ITERS - 1, batch, opt.ns_force_agent, fold_all ? 1 : 0);
        else
            hipLaunchKernelGGL(ns_iter_tail_pair_kernel<false>, ns_tail_grid(NS_PAIR_WG, batch), dim3(256), 4 * 4 * 4 * 64 * sizeof(double), s, A, B, sc,
                               n_sep, NS_ITERS - 1, batch, opt.ns_force_agent, 0);
    }
    if (g_dbg_sigma_stages < 4) return 0;
    if (stream != nullptr && early_logdet && batch == 1) {
        // the noise GEMM rides in the finalize launch (ns_finalize_stream_kernel): one workgroup factors, the others multiply
        // workers b = 1 .. W take the tiles [8 b, 8 b + 8); the tiles [0, 8) are hosted one per worker (ns_finalize_stream_kernel)
        const int ntiles = (stream->N + 31) / 32;
        const int workers = (ntiles - 8 + 7) / 8;
        if (ntiles >= 72 && workers <= 255) {
            hipLaunchKernelGGL(ns_finalize_stream_kernel, dim3(1 + workers), dim3(512), lds, s, Z[0], Z[1], Zt[0], Zt[1], sc, sample_sigma,
                               *stream, status);
            COVO_CHECK_HIP(hipGetLastError());
            if (streamed != nullptr) *streamed = true;
            return 0;
        }
    }
    EpsGenArgs g;
    g.eps_tiled = nullptr;
    g.dyn = nullptr;
    g.sample_offset = 0;
    g.N = 0;
    g.n_inst = 1;
    g.dyn_stride = 0;
    g.eps_stride = 0;
    int passengers = 0;
    if (gen != nullptr && gen->eps_tiled != nullptr) {
        g = *gen;
        const long long ntiles = (long long)((g.N + 31) / 32) * g.n_inst;
        passengers = (int)((ntiles + 7) / 8);  // 8 waves per workgroup, one tile per wave ...
        const int nfac = early_logdet ? batch : 2 * batch;  // factoring workgroups (Z, and B where its log det is not there yet)
        const int room = nfac < 128 ? 256 - nfac : 128;  // ... at most one workgroup on every other CU, waves stride over tiles
        if (passengers > room) passengers = room;
    }
    if (cov != nullptr) {
        std::memset(cov, 0, sizeof(*cov));
        if (batch == 1 && Sigma != nullptr) {  // a_cov is left to the consumer of L (CovDeferred)
            for (int k = 0; k < 2; ++k) {
                cov->Z[k] = Z[k];
                cov->Zt[k] = Zt[k];
            }
            cov->zbuf = sc + SC_ZBUF;
            cov->cz = sc + SC_CZ;
            cov->zcoef = sc + SC_ZCOEF;
            cov->u = sc + SC_U;
            cov->out = Sigma;
            Sigma = nullptr;
        }
    }
    hipLaunchKernelGGL(ns_finalize_kernel, dim3((early_logdet ? batch : 2 * batch) + passengers), dim3(512), lds, s, Z[0], Z[1], Zt[0],
                       Zt[1], sc, sample_sigma, Sigma, L, batch, g, status, A, early_logdet ? 0 : 1);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}
