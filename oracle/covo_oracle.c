/* Plain-C restatement of the hot loops of one quadjax MPC control step.
 *
 * TEST INFRASTRUCTURE ONLY: linked by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py -- never by the product package.
 * PARITY UNPINNED: the JAX reference cannot run in the build container and ships
 * no fixtures; this file follows the cited reference lines and is pinned by the
 * hand-derived known answers in tests/golden/kat.json only.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC -> oracle/_build/libcovo_oracle.so)
 * -ffp-contract=off keeps gcc from fusing a*b+c so the fp32 path rounds like
 * unfused XLA-CPU arithmetic; the fmaf() calls in oracle_noise_gemm_f32 are explicit.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>

#define SUF(x) x##_f32
#define REAL float
#define SQRT sqrtf
#define LOG logf
#define EXP expf
#define ATAN2 atan2f
#define FABS fabsf
#include "covo_oracle_body.h"
#undef SUF
#undef REAL
#undef SQRT
#undef LOG
#undef EXP
#undef ATAN2
#undef FABS

#define SUF(x) x##_f64
#define REAL double
#define SQRT sqrt
#define LOG log
#define EXP exp
#define ATAN2 atan2
#define FABS fabs
#include "covo_oracle_body.h"
#undef SUF
#undef REAL

/* controllers/covo.py:215-224: a = clip(mean + chol(cov) @ eps, -1, 1).
 * jax.random.multivariate_normal(method='cholesky') = mean + einsum(factor, eps).
 * The dot product is an ascending-k fp32 fmaf chain starting from 0 -- exactly
 * what a k-ordered v_mfma_f32_32x32x2_f32 accumulation produces (bit-for-bit).
 * L (n,n) row-major lower-triangular; eps, a (N,n) row-major. */
void oracle_noise_gemm_f32(const float *L, const float *mu, const float *eps, long N, int n, float *a)
{
#pragma omp parallel for schedule(static)
    for (long s = 0; s < N; ++s) {
        const float *e = eps + (size_t)s * n;
        for (int i = 0; i < n; ++i) {
            float acc = 0.0f;
            for (int k = 0; k < n; ++k) acc = fmaf(L[(size_t)i * n + k], e[k], acc);
            float v = mu[i] + acc;
            a[(size_t)s * n + i] = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
        }
    }
}

/* mppi.py:53-66: per-step 4x4 lower factors Ls (H,4,4); eps, a (N,H,4). */
void oracle_noise_blockdiag_f32(const float *Ls, const float *mu, const float *eps, long N, int H, float *a)
{
#pragma omp parallel for schedule(static)
    for (long s = 0; s < N; ++s)
        for (int t = 0; t < H; ++t)
            for (int i = 0; i < 4; ++i) {
                float acc = 0.0f;
                for (int k = 0; k < 4; ++k) acc = fmaf(Ls[(t * 4 + i) * 4 + k], eps[((size_t)s * H + t) * 4 + k], acc);
                float v = mu[t * 4 + i] + acc;
                a[((size_t)s * H + t) * 4 + i] = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
            }
}

/* One whole sampling step (covo.py:212-278 with Sigma's factor given), fp32,
 * used as the timed CPU "port" baseline: noise GEMM -> rollout -> softmax update.
 * work: a (N,128), cost (N).  Returns a_mean_out (n). */
void oracle_sampling_step_f32(const double *prm, int max_steps, const float *state22, int time, const float *pos_traj,
                              const float *vel_traj, int T, const float *L, const float *a_mean, const float *eps, long N,
                              int H, float lam, float gamma_mean, float discount, float *a_work, float *cost_work,
                              float *a_mean_out)
{
    const int n = H * 4;
    const float zero3[3] = {0, 0, 0};
    oracle_noise_gemm_f32(L, a_mean, eps, N, n, a_work);
    oracle_rollout_f32(prm, max_steps, state22, time, pos_traj, vel_traj, T, a_work, N, H, discount, zero3, cost_work, NULL,
                       NULL);
    float m, s;
    float v[512];
    oracle_softmax_partial_f32(cost_work, a_work, N, n, lam, &m, &s, v);
    for (int j = 0; j < n; ++j) a_mean_out[j] = v[j] / s * gamma_mean + a_mean[j] * (1.0f - gamma_mean);
}
