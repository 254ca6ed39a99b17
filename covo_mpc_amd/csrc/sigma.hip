// sigma.hip -- CoVO's optimal sampling covariance and its Cholesky factor (gfx950, fp64, one
// workgroup per matrix; `batch` matrices per launch).
//
// Replaces quadjax/controllers/covo.py:116-132 (optimize_sigma: symmetrise, jnp.linalg.eigh,
// spectrum shift to min 1e-2, Sigma = U diag(exp(log_s)) U^T with det Sigma = sigma^(2n),
// symmetrise) and the Cholesky factorisation jax.random.multivariate_normal applies to a_cov
// (covo.py:216-218; mppi.py:59 for the 4x4 blocks).
//
// Eigendecomposition: one-sided (Hestenes) cyclic Jacobi on G = A + shift*I, where `shift` is a
// Gershgorin bound (+1) that makes G symmetric positive definite with eigenvalues >= 1.  Right
// rotations orthogonalise the columns of G; at convergence G = V diag(lam'), so
//     lam_k = |g_k| - shift,     v_k = g_k / |g_k|,
// and only ONE n x n fp64 matrix (128 KiB of the CU's 160 KiB LDS, column-major) is ever
// stored -- no separate eigenvector matrix.  Sigma = sum_k f(lam_k) v_k v_k^T = H H^T with
// h_k = g_k sqrt(f_k)/|g_k|.  Register-blocked sweep: 32 blocks of 4 columns, block round-robin, 16 cross
// rotations per LDS round trip (see the sweep below); sweeps until no pair exceeds 1e-12 relative.
// Sigma is rounded to fp32 (the reference's a_cov dtype) before its Cholesky factor is taken in
// fp64 and rounded to fp32.  Latency-bound section of covo-online (report us).
#include "covo_common.hpp"
#include "wave_reduce.hpp"
#include "chol_lds.hpp"

constexpr int SG_N = COVO_NA;          // 128
constexpr int SG_LD = SG_N;            // column stride (doubles); bank spreading is done by chunk rotation
constexpr int SG_THREADS = 512;  // 8 waves = 2 per SIMD: up to 256 VGPRs for the register-blocked sweep
constexpr int SG_MAX_SWEEPS = 16;
constexpr double SG_TOL = 1e-12;   // |g_p.g_q| <= tol |g_p||g_q| for every pair of a whole sweep

using wr::group32_allsum;

__device__ __forceinline__ double group16_sum(double v)
{
    // lanes of a pair group are 16 consecutive lanes; xor-butterfly stays inside the group
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// In-LDS lower Cholesky of the n x n SPD matrix stored column-major with stride ld in A
// (right-looking; entries above the diagonal are left untouched).  All threads of the block call.
__device__ void cholesky_lds(double *A, int n, int ld, int tid, int nthreads)
{
    for (int j = 0; j < n; ++j) {
        __syncthreads();
        const double djj = sqrt(A[j * ld + j]);
        const double inv = 1.0 / djj;
        __syncthreads();
        // scale column j
        for (int i = j + tid; i < n; i += nthreads) A[j * ld + i] = (i == j) ? djj : A[j * ld + i] * inv;
        __syncthreads();
        // trailing update: A[i][c] -= L[i][j] L[c][j] for j < c <= i
        const int m = n - j - 1;
        for (int e = tid; e < m * m; e += nthreads) {
            const int c = j + 1 + e / m, i = j + 1 + e % m;
            if (i >= c) A[c * ld + i] -= A[j * ld + i] * A[j * ld + c];
        }
    }
    __syncthreads();
}

// prof (nullable): block 0 / thread 0 stores s_memtime ticks at phase boundaries:
//   [0] start [1] after load+shift [2..2+S] after each sweep ... [20] sweeps done [21] spectrum map
//   [22] H H^T [23] Cholesky done [24] number of sweeps
__global__ __launch_bounds__(SG_THREADS) void sigma_kernel(const double *__restrict__ Rin, float sample_sigma,
                                                           float *__restrict__ Sigma_out, float *__restrict__ L_out,
                                                           unsigned long long *__restrict__ prof)
{
#define SG_PROF(i) do { if (prof && blockIdx.x == 0 && threadIdx.x == 0) prof[i] = __builtin_amdgcn_s_memtime(); } while (0)
    SG_PROF(0);
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *G = sm;                        // [SG_N][SG_LD] column-major
    double *vec = sm + SG_N * SG_LD;       // [SG_N] squared column norms / eigenvalues / scales
    double *red = vec + SG_N;              // [32] reduction scratch
    int *flag = reinterpret_cast<int *>(red + 32);

    const int tid = threadIdx.x;
    const double *__restrict__ R = Rin + (size_t)blockIdx.x * SG_N * SG_N;

    // ---- G = (R + R^T)/2, Gershgorin bound
    for (int e = tid; e < SG_N * SG_N; e += SG_THREADS) {
        const int c = e / SG_N, r = e % SG_N;
        G[c * SG_LD + r] = 0.5 * (R[(size_t)r * SG_N + c] + R[(size_t)c * SG_N + r]);  // covo.py:117
    }
    __syncthreads();
    if (tid < SG_N) {
        double s = 0.0;
        for (int r = 0; r < SG_N; ++r) s += fabs(G[tid * SG_LD + r]);
        vec[tid] = s;
    }
    __syncthreads();
    if (tid == 0) {
        double mx = 0.0;
        for (int i = 0; i < SG_N; ++i) mx = fmax(mx, vec[i]);
        red[0] = mx + 1.0;
    }
    __syncthreads();
    const double shift = red[0];
    if (tid < SG_N) G[tid * SG_LD + tid] += shift;
    __syncthreads();
    SG_PROF(1);

    // ---- one-sided cyclic BLOCK Jacobi.  Columns are grouped in 32 blocks of 4; a block round-robin
    // (31 block-rounds x 16 disjoint block pairs) visits every block pair once per sweep.  A block pair
    // (8 columns) is pulled into the registers of a 32-lane group (lane l owns rows {l + 32 i}), its 16
    // cross rotations are applied back to back, and the 8 columns go back to LDS: one LDS round trip per
    // 16 rotations instead of per rotation (the unblocked sweep was LDS-write-bound).
    // Pairs inside a block are rotated in the first block-round of every sweep.  Squared column norms
    // ride along in registers; only the cross product gamma needs a (32-lane) reduction.  The rotation
    // angle is seeded in fp32 (hardware rcp / sqrt / rsq) and (c, s) are polished in fp64 so that
    // c^2 + s^2 = 1 to rounding: an inexact ANGLE only costs convergence speed, never orthogonality.
    // 512 threads = 16 groups of 32 lanes = the 16 block pairs of a block-round, 2 waves per SIMD.
    const int slot = tid >> 5, l32 = tid & 31;
    // Four column-disjoint rotations (x[a], y[a]), a = 0..3, issued together: branch-free (a pair below
    // the threshold gets the identity rotation) so their latency chains interleave.
    auto rotate4 = [&](double *x0, double *y0, double &ax0, double &ay0, double *x1, double *y1, double &ax1, double &ay1,
                       double *x2, double *y2, double &ax2, double &ay2, double *x3, double *y3, double &ax3,
                       double &ay3) -> bool {
        double *xs[4] = {x0, x1, x2, x3}, *ys[4] = {y0, y1, y2, y3};
        double *als[4] = {&ax0, &ax1, &ax2, &ax3}, *bes[4] = {&ay0, &ay1, &ay2, &ay3};
        double ga[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ga[r] = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) ga[r] = fma(xs[r][i], ys[r][i], ga[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ga[r] = group32_allsum(ga[r]);
        bool any = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double al = *als[r], be = *bes[r];
            const bool live = ga[r] * ga[r] > (SG_TOL * SG_TOL) * al * be;  // uniform within the 32-lane group
            any |= live;
            const float zeta = (float)(be - al) * __builtin_amdgcn_rcpf(2.0f * (float)ga[r]);
            float tf = __builtin_copysignf(1.0f, zeta) *
                       __builtin_amdgcn_rcpf(__builtin_fabsf(zeta) + __builtin_amdgcn_sqrtf(fmaf(zeta, zeta, 1.0f)));
            tf = live ? tf : 0.0f;
            const double t = (double)tf;
            const double u = fma(t, t, 1.0);
            double c = (double)__builtin_amdgcn_rsqf((float)u);
            c = c * fma(-0.5 * u, c * c, 1.5);
            c = c * fma(-0.5 * u, c * c, 1.5);
            const double sn = c * t;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double xi = xs[r][i], yi = ys[r][i];
                xs[r][i] = c * xi - sn * yi;
                ys[r][i] = sn * xi + c * yi;
            }
            const double cc = c * c, ss = sn * sn, cs2 = 2.0 * c * sn * ga[r];
            *als[r] = cc * al - cs2 + ss * be;
            *bes[r] = ss * al + cs2 + cc * be;
        }
        return any;
    };
    constexpr int NB = SG_N / 4;  // 32 column blocks
    for (int sweep = 0; sweep < SG_MAX_SWEEPS; ++sweep) {
        if (tid < SG_N) {  // refresh the squared norms once per sweep (kills the drift of the updates)
            double s2 = 0.0;
            for (int r = 0; r < SG_N; ++r) s2 = fma(G[tid * SG_LD + r], G[tid * SG_LD + r], s2);
            vec[tid] = s2;
        }
        if (tid == 0) *flag = 0;
        __syncthreads();
        bool rotated = false;
        for (int round = 0; round < NB - 1; ++round) {
            {
                int P, Q;
                if (slot == 0) {
                    P = round;
                    Q = NB - 1;
                } else {
                    P = (round + slot) % (NB - 1);
                    Q = (round - slot + (NB - 1)) % (NB - 1);
                }
                double X[4][4], Y[4][4], nx[4], ny[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    nx[a] = vec[4 * P + a];
                    ny[a] = vec[4 * Q + a];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        X[a][i] = G[(4 * P + a) * SG_LD + l32 + 32 * i];
                        Y[a][i] = G[(4 * Q + a) * SG_LD + l32 + 32 * i];
                    }
                }
                if (round == 0) {  // pairs inside each block, once per sweep: 3 steps of 4 disjoint pairs
                    rotated |= rotate4(X[0], X[1], nx[0], nx[1], X[2], X[3], nx[2], nx[3], Y[0], Y[1], ny[0], ny[1], Y[2],
                                       Y[3], ny[2], ny[3]);
                    rotated |= rotate4(X[0], X[2], nx[0], nx[2], X[1], X[3], nx[1], nx[3], Y[0], Y[2], ny[0], ny[2], Y[1],
                                       Y[3], ny[1], ny[3]);
                    rotated |= rotate4(X[0], X[3], nx[0], nx[3], X[1], X[2], nx[1], nx[2], Y[0], Y[3], ny[0], ny[3], Y[1],
                                       Y[2], ny[1], ny[2]);
                }
                // 16 cross pairs as 4 diagonals (a, (a+d)%4) of 4 column-disjoint rotations each
                rotated |= rotate4(X[0], Y[0], nx[0], ny[0], X[1], Y[1], nx[1], ny[1], X[2], Y[2], nx[2], ny[2], X[3], Y[3],
                                   nx[3], ny[3]);
                rotated |= rotate4(X[0], Y[1], nx[0], ny[1], X[1], Y[2], nx[1], ny[2], X[2], Y[3], nx[2], ny[3], X[3], Y[0],
                                   nx[3], ny[0]);
                rotated |= rotate4(X[0], Y[2], nx[0], ny[2], X[1], Y[3], nx[1], ny[3], X[2], Y[0], nx[2], ny[0], X[3], Y[1],
                                   nx[3], ny[1]);
                rotated |= rotate4(X[0], Y[3], nx[0], ny[3], X[1], Y[0], nx[1], ny[0], X[2], Y[1], nx[2], ny[1], X[3], Y[2],
                                   nx[3], ny[2]);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        G[(4 * P + a) * SG_LD + l32 + 32 * i] = X[a][i];
                        G[(4 * Q + a) * SG_LD + l32 + 32 * i] = Y[a][i];
                    }
                    if (l32 == 0) {
                        vec[4 * P + a] = nx[a];
                        vec[4 * Q + a] = ny[a];
                    }
                }
            }
            __syncthreads();
        }
        if (rotated) *flag = 1;
        __syncthreads();
        const int any = *flag;
        __syncthreads();
        SG_PROF(2 + sweep);
        if (prof && blockIdx.x == 0 && tid == 0) prof[24] = sweep + 1;
        if (!any) break;
    }
    SG_PROF(20);

    // ---- eigenvalues (covo.py:118-122) and the spectrum map (covo.py:124-128)
    if (tid < SG_N) {
        double s = 0.0;
        for (int r = 0; r < SG_N; ++r) s = fma(G[tid * SG_LD + r], G[tid * SG_LD + r], s);
        vec[tid] = sqrt(s);  // lam'_k = |g_k|
    }
    __syncthreads();
    if (tid == 0) {
        double mn = vec[0] - shift;
        for (int i = 1; i < SG_N; ++i) mn = fmin(mn, vec[i] - shift);
        red[1] = mn;
    }
    __syncthreads();
    const double min_eign = red[1];
    double log_o = 0.0;
    if (tid < SG_N) {
        const double o = (vec[tid] - shift) + (-min_eign + 1e-2);  // eigns + offset
        log_o = log(o);
    }
    __syncthreads();
    // sum of log_o over the 128 eigenvalues (two waves)
    {
        double v = (tid < SG_N) ? log_o : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (tid < SG_N && (tid & 63) == 0) red[4 + (tid >> 6)] = v;
    }
    __syncthreads();
    if (tid < SG_N) {
        const double sum_log_o = red[4] + red[5];
        const double n = (double)SG_N;
        const double log_det_a_cov = n * (log((double)sample_sigma) * 2.0);
        const double log_const = (log_det_a_cov * 2.0 + sum_log_o) / n;
        const double log_s = 0.5 * log_const - 0.5 * log_o;
        const double f = exp(log_s);
        // h_k = g_k * sqrt(f_k) / |g_k|
        const double sc = sqrt(f) / vec[tid];
        vec[tid] = sc;
    }
    __syncthreads();
    for (int e = tid; e < SG_N * SG_N; e += SG_THREADS) {
        const int c = e / SG_N, r = e % SG_N;
        G[c * SG_LD + r] *= vec[c];
    }
    __syncthreads();

    SG_PROF(21);
    // ---- Sigma = H H^T (covo.py:130-132; H H^T is symmetric by construction): two 4x4 tiles per thread
    double acc[2][4][4];
    int bi[2], bj[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int tile = tid + h * SG_THREADS;  // 1024 tiles of 4x4
        bi[h] = (tile >> 5) * 4;
        bj[h] = (tile & 31) * 4;
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[h][x][y] = 0.0;
    }
    for (int k = 0; k < SG_N; ++k) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double hi[4], hj[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                hi[x] = G[k * SG_LD + bi[h] + x];
                hj[x] = G[k * SG_LD + bj[h] + x];
            }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[h][x][y] = fma(hi[x], hj[y], acc[h][x][y]);
        }
    }
    __syncthreads();
    // round to fp32 (a_cov dtype), publish, and keep the fp32-rounded values for the factorisation
    float *So = Sigma_out ? Sigma_out + (size_t)blockIdx.x * SG_N * SG_N : nullptr;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const float v = (float)acc[h][x][y];
                if (So) So[(size_t)(bi[h] + x) * SG_N + bj[h] + y] = v;
                G[(bj[h] + y) * SG_LD + bi[h] + x] = (double)v;  // column-major: G[col][row]
            }
    __syncthreads();
    // exact symmetry of the fp32 image: (a + a^T)/2 with a already symmetric to rounding
    for (int e = tid; e < SG_N * SG_N; e += SG_THREADS) {
        const int c = e / SG_N, r = e % SG_N;
        if (r > c) {
            const double v = 0.5 * (G[c * SG_LD + r] + G[r * SG_LD + c]);
            G[c * SG_LD + r] = v;
        }
    }
    __syncthreads();

    SG_PROF(22);
    // ---- lower Cholesky factor (covo.py:216 via multivariate_normal)
    chol_lds_fast(G, SG_N, SG_LD, tid, SG_THREADS);
    SG_PROF(23);
    float *Lo = L_out + (size_t)blockIdx.x * SG_N * SG_N;
    for (int e = tid; e < SG_N * SG_N; e += SG_THREADS) {
        const int r = e / SG_N, c = e % SG_N;
        Lo[e] = (c <= r) ? (float)G[c * SG_LD + r] : 0.0f;
    }
}

// Batched lower Cholesky of fp32 SPD matrices (n <= 128), fp64 internally, one workgroup each.
__global__ __launch_bounds__(256) void cholesky_kernel(const float *__restrict__ Ain, int n, float *__restrict__ Lout)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int ld = n + 1;
    const float *A = Ain + (size_t)blockIdx.x * n * n;
    float *L = Lout + (size_t)blockIdx.x * n * n;
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
        const int r = e / n, c = e % n;
        if (r >= c) sm[c * ld + r] = 0.5 * ((double)A[(size_t)r * n + c] + (double)A[(size_t)c * n + r]);
    }
    if ((n & 7) == 0) chol_lds_fast(sm, n, ld, threadIdx.x, blockDim.x);  // blockDim is wave-uniformly 64 or 256
    else cholesky_lds(sm, n, ld, threadIdx.x, blockDim.x);
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
        const int r = e / n, c = e % n;
        L[e] = (c <= r) ? (float)sm[c * ld + r] : 0.0f;
    }
}

int launch_sigma(const double *R, int batch, float sample_sigma, float *Sigma, float *L, unsigned long long *prof,
                 hipStream_t s)
{
    const size_t lds = (size_t)(SG_N * SG_LD + SG_N + 32 + 2) * sizeof(double);
    static unsigned long long attr_devices = 0;  // (per device: covo_first_on_device)
    if (covo_first_on_device(attr_devices)) {
        COVO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sigma_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(sigma_kernel, dim3(batch), dim3(SG_THREADS), lds, s, R, sample_sigma, Sigma, L, prof);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_cholesky(const float *A, int n, int batch, float *L, hipStream_t s)
{
    if (n < 1 || n > 128) {
        covo_set_error("covo_cholesky: n=%d out of range [1,128]", n);
        return COVO_E_BADARG;
    }
    const size_t lds = (size_t)n * (n + 1) * sizeof(double);
    static unsigned long long attr_devices = 0;  // (per device: covo_first_on_device)
    if (covo_first_on_device(attr_devices)) {
        COVO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(cholesky_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 129 * 8));
    }
    hipLaunchKernelGGL(cholesky_kernel, dim3(batch), dim3(n <= 8 ? 64 : 256), lds, s, A, n, L);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}
