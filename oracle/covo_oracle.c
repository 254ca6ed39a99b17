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
#define SIN sinf
#include "covo_oracle_body.h"
#undef SUF
#undef REAL
#undef SQRT
#undef LOG
#undef EXP
#undef ATAN2
#undef FABS
#undef SIN

#define SUF(x) x##_f64
#define REAL double
#define SQRT sqrt
#define LOG log
#define EXP exp
#define ATAN2 atan2
#define FABS fabs
#define SIN sin
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
                       NULL, 0);
    float m, s;
    float v[512];
    oracle_softmax_partial_f32(cost_work, a_work, N, n, lam, &m, &s, v);
    for (int j = 0; j < n; ++j) a_mean_out[j] = v[j] / s * gamma_mean + a_mean[j] * (1.0f - gamma_mean);
}

/* ------------------------------------------------------------------------------------------------
 * controllers/covo.py:134-185: R = jacfwd(jacfwd(get_cumulated_cost))(a_mean), the exact Hessian of
 *   C(a) = -( sum_{k<H} r(s_k) + r(s_0) ),  s_{k+1} = step_env(s_k, a_k, deterministic=True)
 * (no discount, no done-freeze).  Forward-over-forward AD = hyper-dual arithmetic: one rollout per unordered
 * pair (i <= j) with a_i carrying e1 and a_j carrying e2; R_ij = the e1e2 part.  fp64, OpenMP over pairs.
 * JAX AD conventions: |x|' = sign(x) (0 at 0); jnp.clip = min(max(x,lo),hi) whose lax.max/min JVPs split an
 * exact tie 0.5/0.5 -- the action passes two clips (quadrotor.py:223,258).
 * Part of the timed CPU baseline of bench.py (the reference computes this Hessian every covo-online step). */
typedef struct { double v, a, b, ab; } hd_t;
static inline hd_t hd_c(double v) { hd_t r = {v, 0, 0, 0}; return r; }
static inline hd_t hd_add(hd_t x, hd_t y) { hd_t r = {x.v + y.v, x.a + y.a, x.b + y.b, x.ab + y.ab}; return r; }
static inline hd_t hd_sub(hd_t x, hd_t y) { hd_t r = {x.v - y.v, x.a - y.a, x.b - y.b, x.ab - y.ab}; return r; }
static inline hd_t hd_neg(hd_t x) { hd_t r = {-x.v, -x.a, -x.b, -x.ab}; return r; }
static inline hd_t hd_scale(hd_t x, double c) { hd_t r = {x.v * c, x.a * c, x.b * c, x.ab * c}; return r; }
static inline hd_t hd_addc(hd_t x, double c) { x.v += c; return x; }
static inline hd_t hd_mul(hd_t x, hd_t y)
{
    hd_t r = {x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b, x.ab * y.v + x.a * y.b + x.b * y.a + x.v * y.ab};
    return r;
}
static inline hd_t hd_chain(hd_t x, double f, double d1, double d2)
{
    hd_t r = {f, d1 * x.a, d1 * x.b, d1 * x.ab + d2 * x.a * x.b};
    return r;
}
static inline hd_t hd_sqrt(hd_t x) { double s = sqrt(x.v), d1 = 0.5 / s; return hd_chain(x, s, d1, -0.5 * d1 / x.v); }
static inline hd_t hd_recip(hd_t x) { double r = 1.0 / x.v; return hd_chain(x, r, -r * r, 2.0 * r * r * r); }
static inline hd_t hd_log(hd_t x) { double r = 1.0 / x.v; return hd_chain(x, log(x.v), r, -r * r); }
static inline hd_t hd_abs(hd_t x)
{
    double sg = x.v > 0 ? 1.0 : (x.v < 0 ? -1.0 : 0.0);
    hd_t r = {fabs(x.v), sg * x.a, sg * x.b, sg * x.ab};
    return r;
}
static inline hd_t hd_clip(hd_t x, double lo, double hi)
{
    double g = 1.0, v = x.v;
    if (v < lo) { v = lo; g = 0.0; } else if (v == lo) { g = 0.5; }
    if (v > hi) { v = hi; g = 0.0; } else if (v == hi) { g *= 0.5; }
    hd_t r = {v, g * x.a, g * x.b, g * x.ab};
    return r;
}
static inline hd_t hd_atan2(hd_t y, hd_t x)
{
    double den = x.v * x.v + y.v * y.v, fy = x.v / den, fx = -y.v / den;
    double fyy = -2.0 * x.v * y.v / (den * den), fxx = -fyy, fxy = (y.v * y.v - x.v * x.v) / (den * den);
    hd_t r;
    r.v = atan2(y.v, x.v);
    r.a = fy * y.a + fx * x.a;
    r.b = fy * y.b + fx * x.b;
    r.ab = fy * y.ab + fx * x.ab + fyy * y.a * y.b + fxx * x.a * x.b + fxy * (y.a * x.b + x.a * y.b);
    return r;
}

typedef struct { hd_t pos[3], vel[3], quat[4], omega[3], f[3]; } hd_state;

/* dynamics/utils.py:266-274, 285-294 */
static hd_t hd_reward(const hd_state *s, const double *pos_tar, const double *vel_tar)
{
    hd_t ep = hd_c(0), ev = hd_c(0);
    for (int i = 0; i < 3; ++i) {
        hd_t dp = hd_neg(hd_addc(s->pos[i], -pos_tar[i])), dv = hd_neg(hd_addc(s->vel[i], -vel_tar[i]));
        ep = hd_add(ep, hd_mul(dp, dp));
        ev = hd_add(ev, hd_mul(dv, dv));
    }
    hd_t err_pos = hd_sqrt(ep), err_vel = hd_sqrt(ev);
    const hd_t *q = s->quat;
    hd_t yn = hd_scale(hd_add(hd_mul(q[3], q[2]), hd_mul(q[0], q[1])), 2.0);
    hd_t yd = hd_addc(hd_scale(hd_add(hd_mul(q[1], q[1]), hd_mul(q[2], q[2])), -2.0), 1.0);
    hd_t yaw = hd_abs(hd_atan2(yn, yd));
    hd_t l = hd_log(hd_addc(err_pos, 1.0));
    hd_t lp = hd_scale(err_pos, 0.4);
    lp = hd_add(lp, hd_scale(hd_clip(hd_scale(l, 4.0), 0.0, 1.0), 0.4));
    lp = hd_add(lp, hd_scale(hd_clip(hd_scale(l, 8.0), 0.0, 1.0), 0.2));
    lp = hd_add(lp, hd_scale(hd_clip(hd_scale(l, 16.0), 0.0, 1.0), 0.1));
    lp = hd_add(lp, hd_scale(hd_clip(hd_scale(l, 32.0), 0.0, 1.0), 0.1));
    hd_t r = hd_addc(hd_neg(hd_scale(err_vel, 0.05)), 1.3);
    r = hd_sub(r, lp);
    return hd_sub(r, hd_scale(yaw, 0.2));
}

/* dynamics/utils.py:297-313 */
static hd_t hd_reward_realworld(const hd_state *s, const double *pos_tar)
{
    hd_t pe = hd_c(0);
    for (int i = 0; i < 3; ++i) {
        hd_t d = hd_addc(s->pos[i], -pos_tar[i]);
        pe = hd_add(pe, hd_mul(d, d));
    }
    hd_t cost = hd_add(hd_scale(hd_scale(pe, 1.0 / 3.0), 5.0), hd_scale(hd_addc(hd_neg(hd_mul(s->quat[3], s->quat[3])), 1.0), 3.0));
    return hd_neg(hd_scale(cost, 0.02));
}

/* dynamics/free.py:10-58 on hyper-dual numbers: the disturbance of the NEXT step from the PRE-step state (time = its
 * clock); dist = [kind, period, scale, disturb_params[6]], draw[3] = this step's uniform draw (periodic / mixed) */
static void hd_disturb_next(const hd_state *s, int time, const double *dist, const double *draw, hd_t *out)
{
    const int kind = dist ? (int)dist[0] : 0;
    hd_t drag[3], sn[3], per[3];
    for (int i = 0; i < 3; ++i) {
        hd_t rel = hd_addc(s->vel[i], -dist[3 + i] * 0.5);
        drag[i] = hd_scale(hd_mul(rel, hd_abs(rel)), -fabs(dist[2]) / (1.5 * 1.5));
        const double period = dist[3 + i] * (dist[1] / 3) + dist[1];
        sn[i] = hd_c(dist[3 + i] * dist[2] * sin(2.0 * M_PI / period * (double)time + dist[6 + i] * 2.0 * M_PI));
        per[i] = (time % (int)dist[1]) == 0 ? hd_c(draw[i]) : s->f[i];
    }
    for (int i = 0; i < 3; ++i) {
        if (kind == 2) out[i] = per[i];
        else if (kind == 3) out[i] = sn[i];
        else if (kind == 4) out[i] = drag[i];
        else if (kind == 5) out[i] = hd_scale(hd_add(hd_add(drag[i], sn[i]), per[i]), 1.0 / 3.0);
        else out[i] = hd_c(0.0); /* none; gaussian under deterministic=True (quadrotor.py:234) */
    }
}

/* envs/quadrotor.py:250-263 + dynamics/free.py:74-155; prm as in dyn_step above; s->f = disturbance of THIS step */
static void hd_dyn_step(hd_state *s, const hd_t *act, const double *prm)
{
    const hd_t *f = s->f;
    const double max_thrust = prm[0], dt = prm[7], g = prm[8], m = prm[9], ascale = prm[10], alpha = prm[11];
    hd_t a[4], u[4];
    for (int i = 0; i < 4; ++i) a[i] = hd_clip(hd_clip(act[i], -1.0, 1.0), -1.0, 1.0); /* quadrotor.py:223,258 */
    u[0] = hd_scale(hd_scale(hd_addc(a[0], 1.0), 0.5 * max_thrust), ascale);
    for (int i = 0; i < 3; ++i) u[1 + i] = hd_scale(hd_scale(hd_scale(a[1 + i], prm[1 + i]), 1.0 / prm[1 + i] * prm[4 + i]), ascale);
    hd_t n2 = hd_c(0);
    for (int i = 0; i < 4; ++i) n2 = hd_add(n2, hd_mul(s->quat[i], s->quat[i]));
    hd_t rn = hd_recip(hd_sqrt(n2));
    hd_t x = hd_mul(s->quat[0], rn), y = hd_mul(s->quat[1], rn), z = hd_mul(s->quat[2], rn), w = hd_mul(s->quat[3], rn);
    const hd_t *om = s->omega;
    hd_t Qz[3];
    Qz[0] = hd_scale(hd_add(hd_mul(x, z), hd_mul(y, w)), 2.0);
    Qz[1] = hd_scale(hd_sub(hd_mul(y, z), hd_mul(x, w)), 2.0);
    Qz[2] = hd_add(hd_sub(hd_sub(hd_mul(w, w), hd_mul(x, x)), hd_mul(y, y)), hd_mul(z, z));
    hd_t qd[4];
    qd[0] = hd_scale(hd_add(hd_mul(w, om[0]), hd_sub(hd_mul(y, om[2]), hd_mul(z, om[1]))), 0.5);
    qd[1] = hd_scale(hd_add(hd_mul(w, om[1]), hd_sub(hd_mul(z, om[0]), hd_mul(x, om[2]))), 0.5);
    qd[2] = hd_scale(hd_add(hd_mul(w, om[2]), hd_sub(hd_mul(x, om[1]), hd_mul(y, om[0]))), 0.5);
    qd[3] = hd_scale(hd_neg(hd_add(hd_add(hd_mul(x, om[0]), hd_mul(y, om[1])), hd_mul(z, om[2]))), 0.5);
    hd_t qq[4] = {hd_add(x, hd_scale(qd[0], dt)), hd_add(y, hd_scale(qd[1], dt)), hd_add(z, hd_scale(qd[2], dt)),
                  hd_add(w, hd_scale(qd[3], dt))};
    for (int i = 0; i < 3; ++i) {
        hd_t vd = hd_addc(hd_scale(hd_add(hd_mul(Qz[i], u[0]), f[i]), 1.0 / m), i == 2 ? -g : 0.0);
        s->pos[i] = hd_add(s->pos[i], hd_scale(s->vel[i], dt));
        s->vel[i] = hd_add(s->vel[i], hd_scale(vd, dt));
        s->omega[i] = hd_add(hd_scale(om[i], alpha), hd_scale(u[1 + i], 1.0 - alpha));
    }
    hd_t m2 = hd_c(0);
    for (int i = 0; i < 4; ++i) m2 = hd_add(m2, hd_mul(qq[i], qq[i]));
    hd_t rn2 = hd_recip(hd_sqrt(m2));
    for (int i = 0; i < 4; ++i) s->quat[i] = hd_mul(qq[i], rn2);
}

/* state22 = [pos vel quat omega f_disturb pos_tar vel_tar] (fp64); a_mean (H*4); R (H*4, H*4) row-major.
 * reward_kind: 0 penyaw (utils.py:285-294), 1 realworld (:297-313); dist (nullable) = [kind, period, scale, disturb_params[6]],
 * draws (H,3) = the per-step uniform draws of the disturbance model (get_hessian splits its key once per step, covo.py:151) */
/* table (nullable, (H,4) rows {g_k[3], c_k}): the per-step table of include/covo_hip.h (covo_disturb_table) given explicitly --
 * f_k = c_drag drag(vel_{k-1}) + c_k f_{k-1} + g_k with c_drag = 1 (drag), 1/3 (mixed), 0 otherwise -- instead of the model
 * functions: checks a Hessian kernel on exactly the (fp32-rounded) forces it was handed. */
void oracle_hessian_ex_f64(const double *prm, const double *state22, int time, const double *pos_traj, const double *vel_traj,
                           int T, const double *a_mean, int H, double *R, int reward_kind, const double *dist,
                           const double *draws, const double *table)
{
    const int n = H * 4;
    const long npairs = (long)n * (n + 1) / 2;
#pragma omp parallel for schedule(dynamic, 16)
    for (long q = 0; q < npairs; ++q) {
        int i = 0;
        long rem = q;
        while (rem >= n - i) { rem -= n - i; ++i; }
        const int j = i + (int)rem;
        hd_state s;
        for (int c = 0; c < 3; ++c) {
            s.pos[c] = hd_c(state22[c]); s.vel[c] = hd_c(state22[3 + c]); s.omega[c] = hd_c(state22[10 + c]);
            s.f[c] = hd_c(state22[13 + c]);
        }
        for (int c = 0; c < 4; ++c) s.quat[c] = hd_c(state22[6 + c]);
        double pos_tar[3] = {state22[16], state22[17], state22[18]}, vel_tar[3] = {state22[19], state22[20], state22[21]};
        double acc = 0.0;
        for (int k = 0; k < H; ++k) {
            acc += (reward_kind == 1 ? hd_reward_realworld(&s, pos_tar) : hd_reward(&s, pos_tar, vel_tar)).ab; /* covo.py:169-174 */
            hd_t act[4];
            for (int d = 0; d < 4; ++d) {
                const int idx = 4 * k + d;
                hd_t x = {a_mean[idx], idx == i ? 1.0 : 0.0, idx == j ? 1.0 : 0.0, 0.0};
                act[d] = x;
            }
            hd_t fn[3];
            const double zero9[9] = {0, 1, 0, 0, 0, 0, 0, 0, 0};
            if (table) {
                const int kind = dist ? (int)dist[0] : 0;
                const double cd = kind == 4 ? 1.0 : (kind == 5 ? 1.0 / 3.0 : 0.0);
                for (int c = 0; c < 3; ++c) {
                    hd_t f = hd_c(0.0);
                    if (k + 1 < H) {
                        f = hd_addc(hd_scale(s.f[c], table[4 * (k + 1) + 3]), table[4 * (k + 1) + c]);
                        if (cd != 0.0) {
                            hd_t rel = hd_addc(s.vel[c], -dist[3 + c] * 0.5);
                            f = hd_add(f, hd_scale(hd_mul(rel, hd_abs(rel)), cd * (-fabs(dist[2]) / (1.5 * 1.5))));
                        }
                    }
                    fn[c] = f;
                }
            } else
                hd_disturb_next(&s, time + k, dist ? dist : zero9, draws ? draws + 3 * k : zero9 + 3, fn); /* free.py:147 (pre-step state) */
            hd_dyn_step(&s, act, prm);
            for (int c = 0; c < 3; ++c) s.f[c] = fn[c];
            int idx = time + k + 1;
            idx = idx < 0 ? 0 : (idx > T - 1 ? T - 1 : idx);
            for (int c = 0; c < 3; ++c) { pos_tar[c] = pos_traj[3 * idx + c]; vel_tar[c] = vel_traj[3 * idx + c]; }
        }
        R[(size_t)i * n + j] = -acc; /* r(s_0) (covo.py:176-178) is constant */
        R[(size_t)j * n + i] = -acc;
    }
}

void oracle_hessian_f64(const double *prm, const double *state22, int time, const double *pos_traj, const double *vel_traj,
                        int T, const double *a_mean, int H, double *R)
{
    oracle_hessian_ex_f64(prm, state22, time, pos_traj, vel_traj, T, a_mean, H, R, 0, NULL, NULL, NULL);
}
