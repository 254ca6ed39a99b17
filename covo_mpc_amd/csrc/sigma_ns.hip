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
// The same matrix is obtained here from three GEMM-shaped pieces, each a handful of 2-3 us launches of
// 64 independent waves (v_mfma_f64_16x16x4_f64, one 16x16 tile per wave):
//   1. lambda_min:  X <- X^2 / |X|_F^2  (NS_SQUARINGS times) on X0 = gershgorin*I - R drives X to the
//      dominant eigenspace; a Rayleigh-Ritz step on its RITZ largest-diagonal columns gives lambda_min
//      (exact as soon as the eigenvector lies in the span -- robust to near-degenerate bottoms).
//   2. (R + delta I)^(-1/2): coupled Newton-Schulz  T = (3I - ZY)/2, Y <- YT, Z <- TZ  from Y0 = B/s,
//      Z0 = I (quadratically convergent, clustering-insensitive, all iterates are polynomials in B and
//      hence symmetric -- which lets every MFMA operand be read in its coalesced orientation;
//      iterates are kept EXACTLY symmetric by computing lower tiles only and mirroring).
//   3. log det B from a single-workgroup Cholesky of B; the final Cholesky of fp32(Sigma) likewise.
// Agreement with the LAPACK-eigh oracle: ~1e-15 relative (fp64), tests/test_gpu_parity.py.
// `batch` matrices per launch (grid.z): covo-offline's 300-step table, env-batched configs.
#include "covo_common.hpp"
#include "wave_reduce.hpp"
#include "chol_lds.hpp"

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int SN = COVO_NA;  // 128
constexpr int NS_SQUARINGS = 16;   // X^(2^16): a relative gap of 2e-4 between the bottom eigenvalue and the 5th one
                                   // (RITZ = 4 are resolved exactly by the Ritz step) is damped to 1e-11
constexpr int NS_ITERS = 24;       // enough for scale/1e-2 up to ~1e6 (CoVO Hessians: 1e4-2e4 with the Gershgorin
                                   // scale, 17-18 iterations); launches after convergence return at once
constexpr double NS_TOL2 = 1e-10;  // iteration k+1 is skipped once |I - Z_k Y_k|_F^2 < 1e-10: step k itself squares
                                   // that residual to ~1e-20, far below the fp32 rounding Sigma gets anyway
constexpr int RITZ = 4;            // Rayleigh-Ritz block: exact lambda_min for up to 4 near-degenerate bottom eigenvalues

// per-matrix scalar slots (doubles)
enum { SC_NORMSQ = 0 /* ..NS_SQUARINGS */, SC_SHIFT = 24, SC_LMIN = 25, SC_DELTA = 26, SC_SCALE = 27, SC_LOGDET = 28,
       SC_ZBUF = 29, SC_ITERS = 30, SC_XBUF = 31, SC_SQ = 23, SC_ERR = 32 /* ..NS_ITERS */, SC_COUNT = 64 };
static_assert(SC_ERR + NS_ITERS <= SC_COUNT && NS_SQUARINGS + 1 <= SC_SHIFT, "scalar slots");

// ---- one wave = one 16x16 tile of C = At^T . B  (At, B row-major 128x128).
// MFMA f64 16x16x4: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; C/D: col = l&15, row = (l>>4) + 4*reg.
// The A operand is fetched as At[k][i], so BOTH operands are read as 128-B contiguous runs; callers pass
// the stored TRANSPOSE of the left factor (every Newton-Schulz iterate is kept together with its
// transpose, both written from the same registers).  No symmetry is assumed or enforced: forcing
// symmetry (mirroring the lower triangle) makes the coupled iteration blow up after convergence.
__device__ __forceinline__ f64x4 tile_mm(const double *__restrict__ A, const double *__restrict__ B, int ti, int tj, int lane)
{
    const int lo = lane & 15, hi = lane >> 4;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    double a[32], b[32];
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
        a[kk] = A[(size_t)(4 * kk + hi) * SN + 16 * ti + lo];
        b[kk] = B[(size_t)(4 * kk + hi) * SN + 16 * tj + lo];
    }
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc, 0, 0, 0);
    return acc;
}

// The squaring stage (power iteration: self-correcting) works on exactly symmetric matrices: only tiles
// with ti >= tj are computed and every value is stored together with its mirror image (36 tiles).
constexpr int NS_TILES = 36;
__device__ __forceinline__ void tri_tile(int w, int &ti, int &tj)
{
    ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= w) ++ti;
    tj = w - ti * (ti + 1) / 2;
}
__device__ __forceinline__ void store_both(double *__restrict__ O, double *__restrict__ Ot, int row, int col, double v)
{
    O[(size_t)row * SN + col] = v;
    Ot[(size_t)col * SN + row] = v;
}
__device__ __forceinline__ void store_sym(double *__restrict__ O, int row, int col, double v)
{
    if (row >= col) {
        O[(size_t)row * SN + col] = v;
        if (row > col) O[(size_t)col * SN + row] = v;
    }
}

// ---- prep: A = (R + R^T)/2, X0 = shift*I - A with a Gershgorin shift, |X0|_F^2
__global__ __launch_bounds__(512) void ns_prep_kernel(const double *__restrict__ Rin, double *__restrict__ A,
                                                      double *__restrict__ X, double *__restrict__ sc)
{
    __shared__ double red[512];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *R = Rin + (size_t)b * SN * SN;
    A += (size_t)b * SN * SN;
    X += (size_t)b * SN * SN;
    sc += (size_t)b * SC_COUNT;
    // thread (q = tid/128, c = tid%128) owns rows r = q, q+4, ... of column c: coalesced in c
    const int c = tid & (SN - 1), q = tid >> 7;
    double colsum = 0.0;
    for (int r = q; r < SN; r += 4) {
        const double v = 0.5 * (R[(size_t)r * SN + c] + R[(size_t)c * SN + r]);  // covo.py:117
        A[(size_t)r * SN + c] = v;
        colsum += fabs(v);
    }
    red[tid] = colsum;
    __syncthreads();
    if (tid < SN) red[tid] = (red[tid] + red[tid + 128]) + (red[tid + 256] + red[tid + 384]);  // column = row abs-sum
    __syncthreads();
    for (int o = 64; o > 0; o >>= 1) {
        if (tid < o) red[tid] = fmax(red[tid], red[tid + o]);
        __syncthreads();
    }
    const double gersh = red[0];
    __syncthreads();
    double fro = 0.0;
    for (int r = q; r < SN; r += 4) fro = fma(A[(size_t)r * SN + c], A[(size_t)r * SN + c], fro);
    red[tid] = fro;
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    // any upper bound of lambda_max(A) works; the tighter it is the faster the power iteration separates
    const double shift = fmin(gersh, sqrt(red[0])) * (1.0 + 1e-12) + 1e-3;
    __syncthreads();
    double nsq = 0.0;
    for (int r = q; r < SN; r += 4) {
        const double v = ((r == c) ? shift : 0.0) - A[(size_t)r * SN + c];
        X[(size_t)r * SN + c] = v;
        nsq = fma(v, v, nsq);
    }
    red[tid] = nsq;
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid < SC_COUNT) sc[tid] = 0.0;
    __syncthreads();
    if (tid == 0) {
        sc[SC_SHIFT] = shift;
        sc[SC_NORMSQ] = red[0];
    }
}

// ---- squaring: Xout = Xin^2 / |Xin|_F^2, accumulates |Xout|_F^2 into sc[SC_NORMSQ + step + 1]
__global__ __launch_bounds__(256) void ns_square_kernel(const double *__restrict__ Xin, double *__restrict__ Xout,
                                                        double *__restrict__ sc, int step, int xbuf_out)
{
    const int b = blockIdx.z, lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);  // lower tile 0..35
    const double *X = Xin + (size_t)b * SN * SN;
    double *O = Xout + (size_t)b * SN * SN;
    double *s = sc + (size_t)b * SC_COUNT;
    // stationary: |X_k|_F^2 stopped moving (X is a projector onto the dominant eigenspace up to scale)
    if (step >= 2 && fabs(s[SC_NORMSQ + step] - s[SC_NORMSQ + step - 1]) <= 1e-11 * s[SC_NORMSQ + step]) {
        if (blockIdx.x == 0 && threadIdx.x == 0) s[SC_NORMSQ + step + 1] = s[SC_NORMSQ + step];
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        s[SC_XBUF] = (double)xbuf_out;
        s[SC_SQ] = (double)(step + 1);
    }
    int ti, tj;
    tri_tile(w, ti, tj);
    const f64x4 acc = tile_mm(X, X, ti, tj, lane);
    const double inv = 1.0 / s[SC_NORMSQ + step];
    double nsq = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double v = acc[r] * inv;
        const int row = 16 * ti + (lane >> 4) + 4 * r, col = 16 * tj + (lane & 15);
        store_sym(O, row, col, v);
        nsq += (row > col) ? 2.0 * v * v : ((row == col) ? v * v : 0.0);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nsq += __shfl_xor(nsq, o, 64);
    if (lane == 0) atomicAdd(&s[SC_NORMSQ + step + 1], nsq);
}

// ---- Rayleigh-Ritz on the RITZ largest-diagonal columns of X: lambda_min(A); then B = A + delta I,
// Y0 = B/s (s = Gershgorin bound of B), Z0 = I.
__global__ __launch_bounds__(512) void ns_ritz_kernel(const double *__restrict__ Aall, const double *__restrict__ X0all,
                                                      const double *__restrict__ X1all, double *__restrict__ Ball, double *__restrict__ Yall,
                                                      double *__restrict__ Ytall, double *__restrict__ Zall,
                                                      double *__restrict__ Ztall, double *__restrict__ sc)
{
    __shared__ double V[RITZ][SN];
    __shared__ double AV[RITZ][SN];
    __shared__ double H[RITZ][RITZ];
    __shared__ double red[512];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const double *A = Aall + (size_t)b * SN * SN;
    const double *X = ((sc[(size_t)b * SC_COUNT + SC_XBUF] != 0.0) ? X1all : X0all) + (size_t)b * SN * SN;
    double *B = Ball + (size_t)b * SN * SN, *Y = Yall + (size_t)b * SN * SN, *Z = Zall + (size_t)b * SN * SN;
    double *Yt = Ytall + (size_t)b * SN * SN, *Zt = Ztall + (size_t)b * SN * SN;
    double *s = sc + (size_t)b * SC_COUNT;
    if (tid < 64) {
        // ---- wave 0: pick the RITZ largest diagonal entries, orthonormalise those columns (two-pass MGS,
        // everything in registers: lane l owns rows l and l+64; reductions on the VALU)
        double d0 = X[(size_t)lane * SN + lane], d1 = X[(size_t)(lane + 64) * SN + lane + 64];
        double v[RITZ][2];
        int pick_k[RITZ];
#pragma unroll
        for (int k = 0; k < RITZ; ++k) {
            // argmax over 128 values: max via butterfly on (value, index) pairs
            double bv = (d0 >= d1) ? d0 : d1;
            int bi = (d0 >= d1) ? lane : lane + 64;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            pick_k[k] = bi;
            if (bi == lane) d0 = -1e300;
            if (bi == lane + 64) d1 = -1e300;
            v[k][0] = X[(size_t)lane * SN + bi];
            v[k][1] = X[(size_t)(lane + 64) * SN + bi];
        }
#pragma unroll
        for (int k = 0; k < RITZ; ++k) {
#pragma unroll
            for (int pass = 0; pass < 2; ++pass)
#pragma unroll
                for (int j = 0; j < k; ++j) {
                    const double dot = wr::wave64_allsum(fma(v[k][0], v[j][0], v[k][1] * v[j][1]));
                    v[k][0] = fma(-dot, v[j][0], v[k][0]);
                    v[k][1] = fma(-dot, v[j][1], v[k][1]);
                }
            const double nrm = sqrt(wr::wave64_allsum(fma(v[k][0], v[k][0], v[k][1] * v[k][1])));
            if (nrm > 1e-280) {
                const double inv = 1.0 / nrm;
                v[k][0] *= inv;
                v[k][1] *= inv;
            } else {  // column numerically inside the span of the previous ones: any unit vector will do
                v[k][0] = (pick_k[k] == lane) ? 1.0 : 0.0;
                v[k][1] = (pick_k[k] == lane + 64) ? 1.0 : 0.0;
            }
            V[k][lane] = v[k][0];
            V[k][lane + 64] = v[k][1];
        }
    }
    __syncthreads();
    // AV = A V (A symmetric: column reads are coalesced), then H = V^T A V
    {
        const int r = tid & (SN - 1), kq = tid >> 7;  // 4 groups of 128 threads
        for (int k = kq; k < RITZ; k += 4) {
            double a0 = 0.0;
            for (int c = 0; c < SN; ++c) a0 = fma(A[(size_t)c * SN + r], V[k][c], a0);
            AV[k][r] = a0;
        }
    }
    __syncthreads();
    if (tid < RITZ * RITZ) {
        const int i = tid / RITZ, j = tid % RITZ;
        double acc = 0.0;
        for (int r = 0; r < SN; ++r) acc = fma(V[i][r], AV[j][r], acc);
        H[i][j] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        // cyclic Jacobi on the RITZ x RITZ symmetric H (serial, eigenvalues only, exits when diagonal)
        double h[RITZ][RITZ];
        for (int i = 0; i < RITZ; ++i)
            for (int j = 0; j < RITZ; ++j) h[i][j] = 0.5 * (H[i][j] + H[j][i]);
        for (int sweep = 0; sweep < 12; ++sweep) {
            double off = 0.0, dia = 0.0;
            for (int p = 0; p < RITZ; ++p)
                for (int q2 = 0; q2 < RITZ; ++q2) (p == q2 ? dia : off) += h[p][q2] * h[p][q2];
            if (off <= 1e-32 * dia) break;
            for (int p = 0; p < RITZ - 1; ++p)
                for (int q2 = p + 1; q2 < RITZ; ++q2) {
                    if (h[p][q2] * h[p][q2] <= 1e-34 * fabs(h[p][p] * h[q2][q2])) continue;
                    const double zeta = (h[q2][q2] - h[p][p]) / (2.0 * h[p][q2]);
                    const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                    for (int k = 0; k < RITZ; ++k) {  // columns
                        const double hp = h[k][p], hq = h[k][q2];
                        h[k][p] = c * hp - sn * hq;
                        h[k][q2] = sn * hp + c * hq;
                    }
                    for (int k = 0; k < RITZ; ++k) {  // rows
                        const double hp = h[p][k], hq = h[q2][k];
                        h[p][k] = c * hp - sn * hq;
                        h[q2][k] = sn * hp + c * hq;
                    }
                }
        }
        double lmin = h[0][0];
        for (int i = 1; i < RITZ; ++i) lmin = fmin(lmin, h[i][i]);
        s[SC_LMIN] = lmin;
        s[SC_DELTA] = -lmin + 1e-2;  // covo.py:120-122: offset = -min_eign + 1e-2
    }
    __syncthreads();
    const double delta = s[SC_DELTA];
    // Gershgorin bound of B = A + delta I (column abs-sums; thread (q, c) owns rows q, q+4, ...)
    const int c = tid & (SN - 1), q = tid >> 7;
    double colsum = 0.0;
    for (int r = q; r < SN; r += 4) colsum += fabs(A[(size_t)r * SN + c] + ((r == c) ? delta : 0.0));
    red[tid] = colsum;
    __syncthreads();
    if (tid < SN) red[tid] = (red[tid] + red[tid + 128]) + (red[tid + 256] + red[tid + 384]);
    __syncthreads();
    for (int o = 64; o > 0; o >>= 1) {
        if (tid < o) red[tid] = fmax(red[tid], red[tid + o]);
        __syncthreads();
    }
    const double gersh = red[0];
    __syncthreads();
    double fro = 0.0;
    for (int r = q; r < SN; r += 4) {
        const double bv = A[(size_t)r * SN + c] + ((r == c) ? delta : 0.0);
        fro = fma(bv, bv, fro);
    }
    red[tid] = fro;
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const double scale = fmin(gersh, sqrt(red[0])) * (1.0 + 1e-12);  // >= lambda_max(B): eigenvalues of Y0 in (0, 1]
    if (tid == 0) s[SC_SCALE] = scale;
    const double inv = 1.0 / scale;
    for (int r = q; r < SN; r += 4) {
        const size_t e = (size_t)r * SN + c;
        const double bv = A[e] + ((r == c) ? delta : 0.0);  // A is exactly symmetric: Y0^T = Y0
        B[e] = bv;
        Y[e] = bv * inv;
        Yt[e] = bv * inv;
        Z[e] = (r == c) ? 1.0 : 0.0;
        Zt[e] = (r == c) ? 1.0 : 0.0;
    }
}

// ---- Newton-Schulz step, part 1:  T = 1.5 I - 0.5 Z.Y  (64 tiles; T and T^T are stored)
__global__ __launch_bounds__(256) void ns_T_kernel(const double *__restrict__ Yall, const double *__restrict__ Ztall,
                                                   double *__restrict__ Tall, double *__restrict__ Ttall,
                                                   double *__restrict__ sc, int iter)
{
    const int b = blockIdx.z, lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
    double *s = sc + (size_t)b * SC_COUNT;
    if (iter > 0 && s[SC_ERR + iter - 1] < NS_TOL2) return;  // converged: Y, Z are final
    const double *Y = Yall + (size_t)b * SN * SN, *Zt = Ztall + (size_t)b * SN * SN;
    double *T = Tall + (size_t)b * SN * SN, *Tt = Ttall + (size_t)b * SN * SN;
    const int ti = w >> 3, tj = w & 7;
    const f64x4 acc = tile_mm(Zt, Y, ti, tj, lane);  // (Z^T)^T . Y = Z.Y
    double err = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + (lane >> 4) + 4 * r, col = 16 * tj + (lane & 15);
        store_both(T, Tt, row, col, ((row == col) ? 1.5 : 0.0) - 0.5 * acc[r]);
        const double d = acc[r] - ((row == col) ? 1.0 : 0.0);
        err = fma(d, d, err);
    }
    err = wr::wave64_allsum(err);
    if (lane == 0) atomicAdd(&s[SC_ERR + iter], err);  // |Z Y - I|_F^2
}

// ---- part 2:  Y' = Y.T (tiles 0..63),  Z' = T.Z (tiles 64..127); each with its transpose
__global__ __launch_bounds__(256) void ns_YZ_kernel(const double *__restrict__ Ytall, const double *__restrict__ Zall,
                                                    const double *__restrict__ Tall, const double *__restrict__ Ttall,
                                                    double *__restrict__ Yout, double *__restrict__ Ytout,
                                                    double *__restrict__ Zout, double *__restrict__ Ztout,
                                                    double *__restrict__ sc, int iter, int zbuf_out)
{
    const int b = blockIdx.z, lane = threadIdx.x & 63;
    double *s = sc + (size_t)b * SC_COUNT;
    if (iter > 0 && s[SC_ERR + iter - 1] < NS_TOL2) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        s[SC_ZBUF] = (double)zbuf_out;  // which Z buffer holds the newest iterate
        s[SC_ITERS] = (double)(iter + 1);
    }
    int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool isZ = w >= 64;
    w &= 63;
    const size_t off = (size_t)b * SN * SN;
    const int ti = w >> 3, tj = w & 7;
    // Y' = Y.T : left factor Y -> pass Y^T;   Z' = T.Z : left factor T -> pass T^T
    const f64x4 acc = isZ ? tile_mm(Ttall + off, Zall + off, ti, tj, lane) : tile_mm(Ytall + off, Tall + off, ti, tj, lane);
    double *O = (isZ ? Zout : Yout) + off, *Ot = (isZ ? Ztout : Ytout) + off;
#pragma unroll
    for (int r = 0; r < 4; ++r) store_both(O, Ot, 16 * ti + (lane >> 4) + 4 * r, 16 * tj + (lane & 15), acc[r]);
}

// ---- log det B via Cholesky (one workgroup per matrix)
__global__ __launch_bounds__(512) void ns_logdet_kernel(const double *__restrict__ Ball, double *__restrict__ sc)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double red[512];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *B = Ball + (size_t)b * SN * SN;
    for (int e = tid; e < SN * SN; e += 512) sm[(e / SN) * (SN + 1) + (e % SN)] = B[e];  // symmetric: row/col agnostic
    chol_lds_fast(sm, SN, SN + 1, tid, 512);
    red[tid] = (tid < SN) ? log(sm[tid * (SN + 1) + tid]) : 0.0;
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) sc[(size_t)b * SC_COUNT + SC_LOGDET] = 2.0 * red[0];
}

// ---- Sigma = c Z / sqrt(s), symmetrised, rounded to fp32; then L = chol(fp32(Sigma)) in fp64 -> fp32
__global__ __launch_bounds__(512) void ns_finalize_kernel(const double *__restrict__ Z0all, const double *__restrict__ Z1all,
                                                          const double *__restrict__ sc, float sample_sigma,
                                                          float *__restrict__ Sigma_out, float *__restrict__ L_out)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *s = sc + (size_t)b * SC_COUNT;
    const double *Z = ((s[SC_ZBUF] != 0.0) ? Z1all : Z0all) + (size_t)b * SN * SN;
    const double n = (double)SN;
    // log_s = 0.5*log_const - 0.5*log_o with log_const = (2*log_det_a_cov + sum log_o)/n  (covo.py:124-128)
    const double log_c = 0.5 * (2.0 * n * (log((double)sample_sigma) * 2.0) + s[SC_LOGDET]) / n;
    const double cz = exp(log_c) / sqrt(s[SC_SCALE]);
    float *So = Sigma_out ? Sigma_out + (size_t)b * SN * SN : nullptr;
    constexpr int LD = SN + 1;
    for (int e = tid; e < SN * SN; e += 512) {
        const int r = e / SN, c = e % SN;
        const float v = (float)(cz * 0.5 * (Z[e] + Z[(size_t)c * SN + r]));  // covo.py:132 symmetrise, a_cov is fp32
        if (So) So[e] = v;
        sm[c * LD + r] = (double)v;
    }
    chol_lds_fast(sm, SN, LD, tid, 512);
    float *Lo = L_out + (size_t)b * SN * SN;
    for (int e = tid; e < SN * SN; e += 512) {
        const int r = e / SN, c = e % SN;
        Lo[e] = (c <= r) ? (float)sm[c * LD + r] : 0.0f;
    }
}

size_t sigma_ns_workspace_bytes(int batch) { return (size_t)batch * (12 * SN * SN + SC_COUNT) * sizeof(double); }

int launch_sigma_ns(const double *R, int batch, float sample_sigma, float *Sigma, float *L, void *workspace,
                    hipStream_t s, hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join)
{
    double *ws = reinterpret_cast<double *>(workspace);
    const size_t M = (size_t)batch * SN * SN;
    double *A = ws, *X0 = ws + M, *X1 = ws + 2 * M, *B = ws + 3 * M;
    double *Y[2] = {ws + 4 * M, ws + 5 * M}, *Yt[2] = {ws + 6 * M, ws + 7 * M};
    double *Z[2] = {ws + 8 * M, ws + 9 * M}, *Zt[2] = {ws + 10 * M, ws + 11 * M};
    double *T = X0, *Tt = X1;  // the squaring buffers are free once lambda_min is known
    double *sc = ws + 12 * M;
    const size_t lds = (size_t)SN * (SN + 1) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        COVO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ns_logdet_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        COVO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ns_finalize_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(ns_prep_kernel, dim3(batch), dim3(512), 0, s, R, A, X0, sc);
    double *xi = X0, *xo = X1;
    for (int i = 0; i < NS_SQUARINGS; ++i) {
        hipLaunchKernelGGL(ns_square_kernel, dim3(NS_TILES / 4, 1, batch), dim3(256), 0, s, xi, xo, sc, i, (xo == X1) ? 1 : 0);
        double *t = xi; xi = xo; xo = t;
    }
    hipLaunchKernelGGL(ns_ritz_kernel, dim3(batch), dim3(512), 0, s, A, X0, X1, B, Y[0], Yt[0], Z[0], Zt[0], sc);
    // log det B only meets the main line again in the finalize kernel: run its Cholesky beside the
    // Newton-Schulz launches on a forked stream (fork/join by events; also valid under stream capture)
    const bool fork = side != nullptr;
    if (fork) {
        COVO_CHECK_HIP(hipEventRecord(ev_fork, s));
        COVO_CHECK_HIP(hipStreamWaitEvent(side, ev_fork, 0));
    }
    hipLaunchKernelGGL(ns_logdet_kernel, dim3(batch), dim3(512), lds, fork ? side : s, B, sc);
    if (fork) COVO_CHECK_HIP(hipEventRecord(ev_join, side));
    for (int i = 0; i < NS_ITERS; ++i) {
        const int in = i & 1, out = in ^ 1;
        hipLaunchKernelGGL(ns_T_kernel, dim3(16, 1, batch), dim3(256), 0, s, Y[in], Zt[in], T, Tt, sc, i);
        hipLaunchKernelGGL(ns_YZ_kernel, dim3(32, 1, batch), dim3(256), 0, s, Yt[in], Z[in], T, Tt, Y[out], Yt[out], Z[out],
                           Zt[out], sc, i, out);
    }
    if (fork) COVO_CHECK_HIP(hipStreamWaitEvent(s, ev_join, 0));
    hipLaunchKernelGGL(ns_finalize_kernel, dim3(batch), dim3(512), lds, s, Z[0], Z[1], sc, sample_sigma, Sigma, L);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}
