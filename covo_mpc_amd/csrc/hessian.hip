// hessian.hip -- exact Hessian of the CoVO rollout objective by hyper-dual forward AD (gfx950, fp64).
// (covo_hessian_pairs: the independent cross-check of the faster second-order-adjoint version, hessian_adj.hip)
//
// Replaces jax.jacfwd(jax.jacfwd(get_cumulated_cost)) of quadjax/controllers/covo.py:134-185:
//     C(a) = -( sum_{k<H} r(s_k) + r(s_0) ),  s_{k+1} = step_env(s_k, a_k, deterministic=True)
// (no discount, no done-freeze; reward on the PRE-step state, so a_{H-1} never reaches a reward and
// rows/cols 124..127 of R are exactly zero; r(s_0) is constant).  R = d^2 C / da^2, a = flattened
// (H,4) mean actions, index 4t+d.
//
// One lane per unordered pair (i <= j): a_i carries e1, a_j carries e2, R_ij = R_ji = C.e1e2.
// Causality: nothing depends on the seeds before step t_i = i/4, so a wave is given pairs that
// share t_i, runs a plain fp64 primal rollout up to t_i and only then switches to hyper-dual
// arithmetic (quad_model.hpp, the same source the fp32 rollout kernel uses).  144 waves per
// matrix; latency-bound (tens of us), fp64 because Sigma's sensitivity needs R to ~1e-6 absolute
// (SURVEY.md App. B).  `batch` matrices per launch (covo-offline table, env-batched config).
//
// JAX AD conventions mirrored (quad_model.hpp): |x|' = sign(x) with sign(0)=0; jnp.clip =
// min(max(x,lo),hi) whose lax.max/min JVPs split an exact tie 0.5/0.5 -- the action passes TWO
// clips inside the Hessian objective (quadrotor.py:223 and :258); norm'(0) = NaN.
#include "covo_common.hpp"
#include "disturb_model.hpp"

struct HessArgs {
    const float *state;     // [batch][COVO_STATE_FLOATS]
    const float *pos_traj;  // [T][3] shared
    const float *vel_traj;
    const float *a_mean;    // [batch][128]
    double *R;              // [batch][128][128]
    int T;
    qm::Consts<double> c;
    const float4 *f_tab;  // [batch][H] rows {g_k[3], c_k} (disturb.hip) or null: no force after step 0
    int reward;           // COVO_REWARD_*
    double drag_k;        // c_drag * (-|disturb_scale| / 1.5^2): != 0 makes the force part of the differentiated state (free.py:41-56)
    double drag_off[3];   // disturb_params[:3] / 2
    // env-batched step: per-instance constants / trajectories / disturbance parameters (null / 0: the shared ones above)
    const qm::Consts<double> *cs;
    size_t traj_stride;
    const dm::Model *models;
};

__host__ __device__ inline double hs_drag_k(const dm::Model &m)
{
    return -(m.kind == COVO_DISTURB_DRAG ? 1.0 : (m.kind == COVO_DISTURB_MIXED ? 1.0 / 3.0 : 0.0)) * fabs((double)m.scale) / 2.25;
}
struct HsDrag {
    double k, off[3];
};

// The force of step k+1 from the PRE-step state of step k (free.py:147): f' = drag_k rel |rel| + c f + g with row k+1 = {g, c} of
// the table (disturb.hip); S = double (primal prefix) or HD.  Without a table: zero.
template <class S>
__device__ __forceinline__ void hs_next_force(const HessArgs &A, const HsDrag &D, int b, int k, const qm::State<S> &s, S (&f)[3])
{
    if (A.f_tab == nullptr || k + 1 >= COVO_H) {
        f[0] = f[1] = f[2] = S{};
        return;
    }
    const float4 r = A.f_tab[(size_t)b * COVO_H + k + 1];
    const double g[3] = {r.x, r.y, r.z}, c = r.w;
    const S v[3] = {s.vx, s.vy, s.vz};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        S n = f[i] * c + g[i];
        if (D.k != 0.0) n = n + qm::drag_force<S, double>(v[i], D.off[i], D.k);
        f[i] = n;
    }
}

constexpr int HS_TI = COVO_H - 1;  // t_i = 0..30 carry lanes; t_i = 31 rows are exact zeros

__host__ __device__ constexpr int hs_pairs(int ti) { return 4 * (COVO_NA - 4 * ti) - 6; }
__host__ __device__ constexpr int hs_waves(int ti) { return (hs_pairs(ti) + 63) / 64; }
__host__ __device__ constexpr int hs_total_waves()
{
    int s = 0;
    for (int t = 0; t < HS_TI; ++t) s += hs_waves(t);
    return s;
}

template <class S>
__device__ __forceinline__ void hs_targets(const float *__restrict__ st, const HessArgs &A, int b, int time0, int k, double (&tar)[6])
{
    if (k == 0) {
        for (int i = 0; i < 3; ++i) { tar[i] = st[ST_POSTAR + i]; tar[3 + i] = st[ST_VELTAR + i]; }
    } else {
        int idx = time0 + k;
        idx = idx < 0 ? 0 : (idx > A.T - 1 ? A.T - 1 : idx);
        const float *pt = A.pos_traj + (size_t)b * A.traj_stride, *vt = A.vel_traj + (size_t)b * A.traj_stride;
        for (int i = 0; i < 3; ++i) { tar[i] = pt[3 * idx + i]; tar[3 + i] = vt[3 * idx + i]; }
    }
}

__global__ __launch_bounds__(64) void hessian_kernel(const HessArgs A)
{
    // wave -> (t_i, chunk)
    int w = blockIdx.x, ti = 0;
    while (w >= hs_waves(ti)) { w -= hs_waves(ti); ++ti; }
    const int b = blockIdx.y;
    const float *__restrict__ st = A.state + (size_t)b * COVO_STATE_FLOATS;
    const float *__restrict__ am = A.a_mean + (size_t)b * COVO_NA;
    double *__restrict__ R = A.R + (size_t)b * COVO_NA * COVO_NA;

    int q = w * 64 + (int)threadIdx.x;
    const bool active = q < hs_pairs(ti);
    if (!active) q = 0;
    int i = 4 * ti, j;
    {
        int d = 0;
        while (q >= COVO_NA - 4 * ti - d) { q -= COVO_NA - 4 * ti - d; ++d; }
        i = 4 * ti + d;
        j = i + q;
    }
    const qm::Consts<double> c = A.cs ? A.cs[b] : A.c;
    HsDrag D = {A.drag_k, {A.drag_off[0], A.drag_off[1], A.drag_off[2]}};
    if (A.models != nullptr) {
        const dm::Model m = A.models[b];
        D.k = hs_drag_k(m);
        for (int q3 = 0; q3 < 3; ++q3) D.off[q3] = 0.5 * (double)m.dp[q3];
    }
    const int time0 = __float_as_int(st[ST_TIME]);

    // ---- primal prefix: steps 0 .. t_i-1 in plain fp64
    qm::State<double> p;
    p.px = st[ST_POS + 0]; p.py = st[ST_POS + 1]; p.pz = st[ST_POS + 2];
    p.vx = st[ST_VEL + 0]; p.vy = st[ST_VEL + 1]; p.vz = st[ST_VEL + 2];
    p.qx = st[ST_QUAT + 0]; p.qy = st[ST_QUAT + 1]; p.qz = st[ST_QUAT + 2]; p.qw = st[ST_QUAT + 3];
    p.ox = st[ST_OMEGA + 0]; p.oy = st[ST_OMEGA + 1]; p.oz = st[ST_OMEGA + 2];
    double fp[3] = {st[ST_FDIST + 0], st[ST_FDIST + 1], st[ST_FDIST + 2]};  // the force acting during the current step
    for (int k = 0; k < ti; ++k) {
        const double a0 = qm::clip11_((double)am[4 * k + 0]), a1 = qm::clip11_((double)am[4 * k + 1]);
        const double a2 = qm::clip11_((double)am[4 * k + 2]), a3 = qm::clip11_((double)am[4 * k + 3]);
        double fn[3] = {fp[0], fp[1], fp[2]};
        hs_next_force<double>(A, D, b, k, p, fn);  // from the PRE-step state
        qm::dyn_step<double, double>(p, a0, a1, a2, a3, c, fp[0], fp[1], fp[2]);
        fp[0] = fn[0]; fp[1] = fn[1]; fp[2] = fn[2];
    }
    // ---- hyper-dual part: steps t_i .. H-1
    qm::State<qm::HD> s;
    s.px = qm::hd(p.px); s.py = qm::hd(p.py); s.pz = qm::hd(p.pz);
    s.vx = qm::hd(p.vx); s.vy = qm::hd(p.vy); s.vz = qm::hd(p.vz);
    s.qx = qm::hd(p.qx); s.qy = qm::hd(p.qy); s.qz = qm::hd(p.qz); s.qw = qm::hd(p.qw);
    s.ox = qm::hd(p.ox); s.oy = qm::hd(p.oy); s.oz = qm::hd(p.oz);
    qm::HD fh[3] = {qm::hd(fp[0]), qm::hd(fp[1]), qm::hd(fp[2])};
    double acc = 0.0;
    for (int k = ti; k < COVO_H; ++k) {
        if (k > ti) {  // s_k depends on the seeds only for k > t_i
            double tar[6];
            hs_targets<double>(st, A, b, time0, k, tar);
            const qm::HD r = qm::reward_kind<qm::HD, double>(A.reward, s, tar[0], tar[1], tar[2], tar[3], tar[4], tar[5]);
            acc += r.ab;
        }
        if (k == COVO_H - 1) break;
        qm::HD a[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int idx = 4 * k + d;
            qm::HD x{(double)am[idx], idx == i ? 1.0 : 0.0, idx == j ? 1.0 : 0.0, 0.0};
            a[d] = qm::clip11_(qm::clip11_(x));  // quadrotor.py:223 and :258
        }
        qm::HD fn[3] = {fh[0], fh[1], fh[2]};
        hs_next_force<qm::HD>(A, D, b, k, s, fn);  // from the PRE-step state (free.py:147)
        qm::dyn_step<qm::HD, double, true, qm::HD>(s, a[0], a[1], a[2], a[3], c, fh[0], fh[1], fh[2]);
        fh[0] = fn[0]; fh[1] = fn[1]; fh[2] = fn[2];
    }
    if (active) {
        R[(size_t)i * COVO_NA + j] = -acc;
        R[(size_t)j * COVO_NA + i] = -acc;
    }
}

int launch_hessian_pairs(const float *state, const float *pos_traj, const float *vel_traj, int T, const covo_env_params &p,
                   const float *a_mean, int batch, double *R, hipStream_t s, const float *f_tab, const void *consts_dev,
                   size_t traj_stride, const void *models_dev)
{
    const bool needs_tab = p.disturb_kind >= COVO_DISTURB_PERIODIC && p.disturb_kind <= COVO_DISTURB_MIXED;
    if (needs_tab && f_tab == nullptr) {
        covo_set_error("hessian_pairs: disturb_kind=%d needs the per-step disturbance table (covo_disturb_table)", p.disturb_kind);
        return COVO_E_BADARG;
    }
    HessArgs A;
    A.state = state;
    A.pos_traj = pos_traj;
    A.vel_traj = vel_traj;
    A.a_mean = a_mean;
    A.R = R;
    A.T = T;
    A.c = make_consts<double>(p);
    const dm::Model m = dm::make_model(p);
    A.f_tab = needs_tab ? reinterpret_cast<const float4 *>(f_tab) : nullptr;
    A.reward = p.reward_kind;
    A.drag_k = hs_drag_k(m);
    for (int i = 0; i < 3; ++i) A.drag_off[i] = 0.5 * (double)m.dp[i];
    A.cs = reinterpret_cast<const qm::Consts<double> *>(consts_dev);
    A.traj_stride = traj_stride;
    A.models = reinterpret_cast<const dm::Model *>(models_dev);
    COVO_CHECK_HIP(hipMemsetAsync(R, 0, (size_t)batch * COVO_NA * COVO_NA * sizeof(double), s));
    hipLaunchKernelGGL(hessian_kernel, dim3(hs_total_waves(), batch), dim3(64), 0, s, A);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}
