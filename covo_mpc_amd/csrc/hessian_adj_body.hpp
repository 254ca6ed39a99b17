// hessian_adj_body.hpp -- the four kernels of the second-order adjoint (see hessian_adj.hip), included TWICE by hessian_adj.hip:
//   ADJ_NX 13, ADJ_FS 0 (namespace adj13): the force of a step is a constant of the step (none / gaussian / periodic / sin);
//   ADJ_NX 16, ADJ_FS 1 (namespace adj16): the force is part of the differentiated state (drag / mixed: f_{k+1} depends on the
//     PRE-step velocity and, for mixed, on f_k; free.py:41-56,147) -- x = (pos, vel, quat, omega, f) fills the 16 rows the tiles
//     were padded to anyway, z = (x, u) has 20 entries, 210 pairs per step.
// No include guard on purpose.
constexpr int NX = ADJ_NX, NZ = NX + 4, NH = NX + 1, HH = COVO_H, NA = COVO_NA;  // state (13, or 16 with the force), step inputs, 1 + NX
// per-matrix workspace (doubles)
constexpr size_t WS_X = 0;                       // [32][16]       x_k
constexpr size_t WS_JF = WS_X + 32 * 16;         // [32][NX][NZ]   df_k/dz (k = 31 unused)
constexpr size_t WS_GL = WS_JF + 32 * NX * NZ;   // [32][16]       grad r_k (k = 0: zeros)
constexpr size_t WS_LAM = WS_GL + 32 * 16;       // [33][16]       lam_k (lam_32 = 0)
constexpr size_t WS_MXX = WS_LAM + 33 * 16;      // [32][16][16]   zero padded
constexpr size_t WS_MXU = WS_MXX + 32 * 256;     // [32][16][4]
constexpr size_t WS_MUU = WS_MXU + 32 * 64;      // [32][4][4]
constexpr size_t WS_S = WS_MUU + 32 * 16;        // [32][16][128]  rows NX..15 zero
constexpr int NPAIR = NZ * (NZ + 1) / 2;         // 153 (210) pairs a <= b of step inputs
constexpr int PAIR_THREADS = (NPAIR + 63) / 64 * 64;
constexpr size_t WS_COUNT = WS_S + (size_t)32 * 16 * NA + 2;
// what KB leaves in WS_LAM before KC runs: the hyper-dual workgroups of KC's launch contract their second derivatives with lam_{k+1}
// as soon as the costate chain of the SAME launch has stored it, and recognise "not yet" by this pattern (a NaN payload no arithmetic
// produces; a NaN costate -- norm'(0) -- is a different bit pattern and passes through)
constexpr unsigned long long ADJ_LAM_UNSET = 0xFFF7ADEADBEEF000ull;


// the disturbance force acting during step k (free.py:91,98,147): the state's own for k = 0, then the table's row k (periodic /
// sin resolved per step by disturb.hip; none / deterministic gaussian: no table, zero)
__device__ __forceinline__ void adj_force(const float *__restrict__ st, const AdjArgs &A, int bi, int k, double (&f)[3])
{
    if (k == 0) {
        f[0] = st[ST_FDIST + 0]; f[1] = st[ST_FDIST + 1]; f[2] = st[ST_FDIST + 2];
    } else if (A.f_tab != nullptr) {
        const float4 r = A.f_tab[(size_t)bi * HH + k];
        f[0] = r.x; f[1] = r.y; f[2] = r.z;
    } else {
        f[0] = f[1] = f[2] = 0.0;
    }
}

__device__ __forceinline__ void adj_targets(const float *__restrict__ st, const float *__restrict__ pos_traj,
                                            const float *__restrict__ vel_traj, int T, int time0, int k, double (&tar)[6])
{
    if (k == 0) {
        for (int i = 0; i < 3; ++i) { tar[i] = st[ST_POSTAR + i]; tar[3 + i] = st[ST_VELTAR + i]; }
    } else {
        int idx = time0 + k;
        idx = idx < 0 ? 0 : (idx > T - 1 ? T - 1 : idx);
        for (int i = 0; i < 3; ++i) { tar[i] = pos_traj[3 * idx + i]; tar[3 + i] = vel_traj[3 * idx + i]; }
    }
}

// the step's 13 outputs / inputs in z order
#define ADJ_FOR_STATE(OP) OP(px, 0) OP(py, 1) OP(pz, 2) OP(vx, 3) OP(vy, 4) OP(vz, 5) OP(qx, 6) OP(qy, 7) OP(qz, 8) OP(qw, 9) OP(ox, 10) OP(oy, 11) OP(oz, 12)

__device__ __forceinline__ void adj_store_state(const qm::State<double> &p, double *__restrict__ x)
{
#define OP(m, i) x[i] = p.m;
    ADJ_FOR_STATE(OP)
#undef OP
}
__device__ __forceinline__ void adj_load_state(qm::State<double> &p, const double *__restrict__ x)
{
#define OP(m, i) p.m = x[i];
    ADJ_FOR_STATE(OP)
#undef OP
}

#if ADJ_FS
// f_{k+1} from the PRE-step state of step k (free.py:147): drag_k rel |rel| + c f_k + g with row k+1 = {g, c} of the table
// (disturb.hip); S = double (primal prefix), D1 or HD.  Beyond the horizon: zero (never integrated).
struct AdjDrag {
    double k, off[3];
};
__device__ __forceinline__ AdjDrag adj_drag(const AdjArgs &A, int bi)
{
    AdjDrag D = {A.drag_k, {A.drag_off[0], A.drag_off[1], A.drag_off[2]}};
    if (A.models != nullptr) {
        const dm::Model m = A.models[bi];
        D.k = adj_drag_coeff(m);
        for (int i = 0; i < 3; ++i) D.off[i] = 0.5 * (double)m.dp[i];
    }
    return D;
}
template <class S>
__device__ __forceinline__ void adj_next_force(const AdjArgs &A, const AdjDrag &D, int bi, int k, const qm::State<S> &s, const S (&f)[3],
                                               S (&fn)[3])
{
    if (k + 1 >= HH) {
        fn[0] = fn[1] = fn[2] = S{};
        return;
    }
    const float4 r = A.f_tab[(size_t)bi * HH + k + 1];
    const double g[3] = {r.x, r.y, r.z}, c = r.w;
    const S v[3] = {s.vx, s.vy, s.vz};
#pragma unroll
    for (int i = 0; i < 3; ++i) fn[i] = (f[i] * c + g[i]) + qm::drag_force<S, double>(v[i], D.off[i], D.k);
}
#endif


#if !ADJ_FS
// ---- KB's primal prefix as parallel scans (round 5).  x_k by a sequential fp64 rollout was the launch's critical path (wave 31: 31
// steps of ~52 dependent-issue instructions, 4.7 us).  With the force of a step a constant of that step (none / gaussian / periodic /
// sin) every recurrence of the step is ASSOCIATIVE over the horizon, so lane k of one wave gets x_k in five scan levels:
//   body rate   w_(k+1) = alpha w_k + (1 - alpha) wt_k           an affine recurrence with a constant factor: a weighted prefix sum;
//   attitude    q_(k+1) = normalise(q_k (x) (dt/2 w_k, 1))        (free.py:96,104,139: q + dt/2 L(q) H w IS that quaternion product, and
//               the re-normalisations are scalars, which commute): q_k = normalise(q_0 (x) r_0 (x) ... (x) r_(k-1)), a prefix PRODUCT;
//   velocity    v_(k+1) = v_k + dt vdot(q_k, thrust_k, f_k), position p_(k+1) = p_k + dt v_k: two prefix sums.
// Every wave of the launch runs the same scan and takes lane k's state.  The operation ORDER differs from the sequential rollout
// (fp64: ~1e-15, against a 1e-9 bar on the Hessian); drag / mixed (the next force depends on the velocity) keep the sequential prefix.
struct AdjQuat {
    double x, y, z, w;
};
__device__ __forceinline__ AdjQuat adj_qmul(const AdjQuat &a, const AdjQuat &b)  // Hamilton product, (x, y, z) vector part, w scalar
{
    AdjQuat r;
    r.x = fma(a.w, b.x, fma(a.x, b.w, fma(a.y, b.z, -(a.z * b.y))));
    r.y = fma(a.w, b.y, fma(a.y, b.w, fma(a.z, b.x, -(a.x * b.z))));
    r.z = fma(a.w, b.z, fma(a.z, b.w, fma(a.x, b.y, -(a.y * b.x))));
    r.w = fma(a.w, b.w, -fma(a.x, b.x, fma(a.y, b.y, a.z * b.z)));
    return r;
}
template <int CTRL>
__device__ __forceinline__ AdjQuat adj_qdpp(const AdjQuat &q)
{
    return AdjQuat{wr::dpp_f64<CTRL>(q.x), wr::dpp_f64<CTRL>(q.y), wr::dpp_f64<CTRL>(q.z), wr::dpp_f64<CTRL>(q.w)};
}
// inclusive prefix sum over lanes 0 .. 31 of v_j weighted by fac^(k - j) (fac = 1: a plain sum).  pw[i] = fac^(2^i), i = 0 .. 3;
// pr = fac^((lane & 15) + 1) (what row 0's total is worth in row 1).  DPP row shifts return 0 where there is no source lane.
__device__ __forceinline__ double adj_scan_sum(double v, const double (&pw)[4], double pr, int lane)
{
    v = fma(wr::dpp_f64<0x111>(v), pw[0], v);   // row_shr:1
    v = fma(wr::dpp_f64<0x112>(v), pw[1], v);   // row_shr:2
    v = fma(wr::dpp_f64<0x114>(v), pw[2], v);   // row_shr:4
    v = fma(wr::dpp_f64<0x118>(v), pw[3], v);   // row_shr:8
    const double t = wr::dpp_f64<0x142>(v);     // row_bcast:15 -- lane 15 of the previous row
    return ((lane & 16) != 0) ? fma(t, pr, v) : v;
}
// -> lane k (< 32): the state BEFORE step k (what `for t < k: dyn_core` leaves), p0 = the state before step 0
template <bool FT>
__device__ __forceinline__ qm::State<double> adj_prefix_scan(const qm::State<double> &p0, const double (*sact)[4], const double (*sfd)[3],
                                                             const qm::Consts<double> &c, int lane)
{
    const int k = lane & 31, lo = lane & 15;
    const double one[4] = {1.0, 1.0, 1.0, 1.0};
    // powers of alpha: alpha^(2^i), alpha^(lo + 1), alpha^k
    double pa[4];
    pa[0] = c.alpha;
    pa[1] = pa[0] * pa[0];
    pa[2] = pa[1] * pa[1];
    pa[3] = pa[2] * pa[2];
    const double a16 = pa[3] * pa[3];
    double alo = 1.0;  // alpha^lo
    if (lo & 1) alo *= pa[0];
    if (lo & 2) alo *= pa[1];
    if (lo & 4) alo *= pa[2];
    if (lo & 8) alo *= pa[3];
    const double ak = (k & 16) ? alo * a16 : alo, apr = alo * c.alpha;
    // body rates: w_k = alpha^k w_0 + sum_(j < k) alpha^(k - 1 - j) (1 - alpha) wt_j
    const double th = sact[k][0];
    double cx = sact[k][1] * c.one_m_alpha, cy = sact[k][2] * c.one_m_alpha, cz = sact[k][3] * c.one_m_alpha;
    cx = adj_scan_sum(cx, pa, apr, lane);
    cy = adj_scan_sum(cy, pa, apr, lane);
    cz = adj_scan_sum(cz, pa, apr, lane);
    const double ox = fma(ak, p0.ox, wr::dpp_f64<0x138>(cx));  // wave_shr:1: the exclusive sum (lane 0: 0)
    const double oy = fma(ak, p0.oy, wr::dpp_f64<0x138>(cy));
    const double oz = fma(ak, p0.oz, wr::dpp_f64<0x138>(cz));
    // attitude: inclusive prefix product of r_j = (dt/2 w_j, 1), then q_k = normalise(q0 (x) r_0 ... r_(k-1))
    AdjQuat P{ox * c.half_dt, oy * c.half_dt, oz * c.half_dt, 1.0};
    {
        AdjQuat t = adj_qdpp<0x111>(P);
        if (lo >= 1) P = adj_qmul(t, P);
        t = adj_qdpp<0x112>(P);
        if (lo >= 2) P = adj_qmul(t, P);
        t = adj_qdpp<0x114>(P);
        if (lo >= 4) P = adj_qmul(t, P);
        t = adj_qdpp<0x118>(P);
        if (lo >= 8) P = adj_qmul(t, P);
        t = adj_qdpp<0x142>(P);
        if (lane & 16) P = adj_qmul(t, P);
    }
    AdjQuat E = adj_qdpp<0x138>(P);  // exclusive
    if (lane == 0) E = AdjQuat{0.0, 0.0, 0.0, 1.0};
    AdjQuat q0{p0.qx, p0.qy, p0.qz, p0.qw};
    {
        const double rn = qm::rsq64_(fma(q0.x, q0.x, fma(q0.y, q0.y, fma(q0.z, q0.z, q0.w * q0.w))));  // free.py:88 (the noisy state is not unit)
        q0.x *= rn; q0.y *= rn; q0.z *= rn; q0.w *= rn;
    }
    AdjQuat q = adj_qmul(q0, E);
    if (k > 0) {  // (step 0's state is the stored one, un-normalised: dyn_core normalises it at entry itself)
        const double rn = qm::rsq64_(fma(q.x, q.x, fma(q.y, q.y, fma(q.z, q.z, q.w * q.w))));
        q.x *= rn; q.y *= rn; q.z *= rn; q.w *= rn;
    }
    // velocity increments of step k from the UNIT attitude of step k (free.py:97-99,103), then the two prefix sums
    const double fx = (FT || k == 0) ? sfd[k][0] : 0.0, fy = (FT || k == 0) ? sfd[k][1] : 0.0, fz = (FT || k == 0) ? sfd[k][2] : 0.0;
    const double Qz0 = 2.0 * (q.x * q.z + q.y * q.w), Qz1 = 2.0 * (q.y * q.z - q.x * q.w), Qz2 = q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z;
    double dvx = ((Qz0 * th + fx) * c.inv_m) * c.dt, dvy = ((Qz1 * th + fy) * c.inv_m) * c.dt, dvz = ((Qz2 * th + fz) * c.inv_m + c.neg_g) * c.dt;
    dvx = adj_scan_sum(dvx, one, 1.0, lane);
    dvy = adj_scan_sum(dvy, one, 1.0, lane);
    dvz = adj_scan_sum(dvz, one, 1.0, lane);
    const double vx = p0.vx + wr::dpp_f64<0x138>(dvx), vy = p0.vy + wr::dpp_f64<0x138>(dvy), vz = p0.vz + wr::dpp_f64<0x138>(dvz);
    double dpx = adj_scan_sum(vx * c.dt, one, 1.0, lane), dpy = adj_scan_sum(vy * c.dt, one, 1.0, lane), dpz = adj_scan_sum(vz * c.dt, one, 1.0, lane);
    qm::State<double> r;
    r.px = p0.px + wr::dpp_f64<0x138>(dpx);
    r.py = p0.py + wr::dpp_f64<0x138>(dpy);
    r.pz = p0.pz + wr::dpp_f64<0x138>(dpz);
    r.vx = vx; r.vy = vy; r.vz = vz;
    r.qx = (k == 0) ? p0.qx : q.x; r.qy = (k == 0) ? p0.qy : q.y; r.qz = (k == 0) ? p0.qz : q.z; r.qw = (k == 0) ? p0.qw : q.w;
    r.ox = ox; r.oy = oy; r.oz = oz;
    return r;
}
#endif

// one step on hyper-dual numbers: z_a carries e1, z_b carries e2 (an index outside 0..16 seeds nothing).
// r = r_k(x) (0 for k = 0), s = f_k(z) (left at x_k for k = H-1, whose dynamics never reach a reward).
__device__ __forceinline__ void adj_hd_step(const float *__restrict__ st, const float *__restrict__ am, const AdjArgs &A, int bi,
                                            int time0, int k, const qm::State<double> &p, int a, int b, qm::HD &r,
                                            qm::State<qm::HD> &s
#if ADJ_FS
                                            , const double (&pf)[3], qm::HD (&fn)[3]  // the force during step k; out: f_{k+1}
#endif
)
{
    const qm::Consts<double> c = A.cs ? A.cs[bi] : A.c;
#define OP(m, i) s.m = qm::HD{p.m, a == i ? 1.0 : 0.0, b == i ? 1.0 : 0.0, 0.0};
    ADJ_FOR_STATE(OP)
#undef OP
#if ADJ_FS
    qm::HD fh[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        fh[i] = qm::HD{pf[i], a == 13 + i ? 1.0 : 0.0, b == 13 + i ? 1.0 : 0.0, 0.0};
        fn[i] = fh[i];
    }
#endif
    r = qm::hd(0.0);
    if (k >= 1) {
        double tar[6];
        adj_targets(st, A.pos_traj + bi * A.traj_stride, A.vel_traj + bi * A.traj_stride, A.T, time0, k, tar);
        r = qm::reward_kind<qm::HD, double>(A.reward, s, tar[0], tar[1], tar[2], tar[3], tar[4], tar[5]);
    }
    if (k <= HH - 2) {
        qm::HD act[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const qm::HD x{(double)am[4 * k + d], a == NX + d ? 1.0 : 0.0, b == NX + d ? 1.0 : 0.0, 0.0};
            act[d] = qm::clip11_(qm::clip11_(x));  // quadrotor.py:223 and :258
        }
#if ADJ_FS
        adj_next_force<qm::HD>(A, adj_drag(A, bi), bi, k, s, fh, fn);  // from the PRE-step state
        qm::dyn_step<qm::HD, double, true, qm::HD>(s, act[0], act[1], act[2], act[3], c, fh[0], fh[1], fh[2]);
#else
        double f[3];
        adj_force(st, A, bi, k, f);
        qm::dyn_step<qm::HD, double>(s, act[0], act[1], act[2], act[3], c, f[0], f[1], f[2]);
#endif
    }
}

// ---- KB: wave k: x_k by a plain fp64 rollout, then df_k/dz (13 x 17) and grad r_k from 17 first-order seeds
// FT: a per-step force table is present (periodic / sin); without it the force of steps >= 1 is a literal zero and its three
// adds per step drop out of the prefix's serial chain
template <bool FT>
__global__ __launch_bounds__(64) void adj_jac_kernel(const AdjArgs A)
{
    const int k = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const float *__restrict__ st = A.state + (size_t)b * COVO_STATE_FLOATS;
    // the mean this step plans around: the shifted one from memory, or (begin work folded in) the caller's unshifted one read through
    // the shift's index map -- entry i of the shifted mean is entry i + 4 of the old one, the last four repeat (covo.py:201-203)
    const bool raw = A.a_mean_raw != nullptr;
    const float *__restrict__ am_base = raw ? A.a_mean_raw : A.a_mean + (size_t)b * NA;
    auto am = [&](int i) { return am_base[(raw && i < NA - 4) ? i + 4 : i]; };
    if (raw && k == 0) {  // what step_begin_kernel leaves for the launches behind this one
        float *shift_out = const_cast<float *>(A.a_mean);
        shift_out[lane] = am(lane);
        shift_out[lane + 64] = am(lane + 64);
        if (lane < 4) step_begin_derive(lane, A.blk, A.derive_keys, A.shared_noise_scale, A.dyn_out);
        if (lane == 4 && A.seq != nullptr) A.seq[0] = A.seq[0] + 1u;
    }
    double *__restrict__ ws = A.ws + (size_t)b * WS_COUNT;
    if (lane < 16) {  // rows k (and 32, by the last wave) of the costate buffer: "not yet stored" for KC's launch (ADJ_LAM_UNSET)
        reinterpret_cast<unsigned long long *>(ws + WS_LAM)[16 * k + lane] = ADJ_LAM_UNSET;
        if (k == HH - 1) reinterpret_cast<unsigned long long *>(ws + WS_LAM)[16 * HH + lane] = ADJ_LAM_UNSET;
    }
    const int time0 = __float_as_int(st[ST_TIME]);
    const qm::Consts<double> c = A.cs ? A.cs[b] : A.c;
    qm::State<double> p;
    p.px = st[ST_POS + 0]; p.py = st[ST_POS + 1]; p.pz = st[ST_POS + 2];
    p.vx = st[ST_VEL + 0]; p.vy = st[ST_VEL + 1]; p.vz = st[ST_VEL + 2];
    p.qx = st[ST_QUAT + 0]; p.qy = st[ST_QUAT + 1]; p.qz = st[ST_QUAT + 2]; p.qw = st[ST_QUAT + 3];
    p.ox = st[ST_OMEGA + 0]; p.oy = st[ST_OMEGA + 1]; p.oz = st[ST_OMEGA + 2];
    // The prefix is the launch's critical path: for wave 31, 31 steps of dependent-issue fp64 instructions, one per ~7 cycles
    // whatever their dependencies.  What does not depend on the state -- clip, thrust and body-rate targets of every step's
    // action (quadrotor.py:223,258-260; 16 of a step's 80 instructions) -- is computed by lane t for step t in ONE pass and
    // read back per step (uniform LDS reads); from step 1 on the entry normalisation of the just-normalised quaternion is
    // skipped (<= 1 ulp; the dual step below keeps both).  80 -> 52 instructions per step, 10.1 -> 7.8 us.  (Measured and
    // dropped: the attitude / translation cascade as TWO waves through LDS, as in the rollout -- the three LDS writes per step
    // on the attitude wave cost more issue time than the 13 instructions they move away: 9.3 us.)
    __shared__ double sact[HH][4];
    __shared__ double sfd[HH][3];  // the force of every step (adj_force), read back per step like the action terms
    if (lane < HH) {
        double f[3];
        adj_force(st, A, b, lane, f);
        sfd[lane][0] = f[0]; sfd[lane][1] = f[1]; sfd[lane][2] = f[2];
    }
    if (lane < HH) {
        const double a0 = qm::clip11_((double)am(4 * lane + 0)), a1 = qm::clip11_((double)am(4 * lane + 1));
        const double a2 = qm::clip11_((double)am(4 * lane + 2)), a3 = qm::clip11_((double)am(4 * lane + 3));
        sact[lane][0] = (a0 + 1.0) * c.thrust_half;
        sact[lane][1] = a1 * c.komega[0];
        sact[lane][2] = a2 * c.komega[1];
        sact[lane][3] = a3 * c.komega[2];
    }
    __syncthreads();
#if ADJ_FS
    const AdjDrag D = adj_drag(A, b);
    double pf[3] = {sfd[0][0], sfd[0][1], sfd[0][2]};  // the force acting during the current step
    for (int t = 0; t < k; ++t) {
        const double th = sact[t][0], w0 = sact[t][1], w1 = sact[t][2], w2 = sact[t][3];
        double fn[3];
        adj_next_force<double>(A, D, b, t, p, pf, fn);  // from the PRE-step state
        if (t == 0) qm::dyn_core<double, double>(p, th, w0, w1, w2, c, pf[0], pf[1], pf[2]);
        else qm::dyn_core<double, double, false>(p, th, w0, w1, w2, c, pf[0], pf[1], pf[2]);
        pf[0] = fn[0]; pf[1] = fn[1]; pf[2] = fn[2];
    }
    if (lane == 0) {
        adj_store_state(p, ws + WS_X + 16 * k);
        ws[WS_X + 16 * k + 13] = pf[0]; ws[WS_X + 16 * k + 14] = pf[1]; ws[WS_X + 16 * k + 15] = pf[2];
    }
#else
    if (A.scan_prefix) {
        // x_k as lane k of five-level scans (adj_prefix_scan) instead of k sequential steps
        const qm::State<double> ps = adj_prefix_scan<FT>(p, sact, sfd, c, lane);
#define OP(m, i) p.m = wr::bcast_lane(ps.m, k);
        ADJ_FOR_STATE(OP)
#undef OP
    } else {
        for (int t = 0; t < k; ++t) {
            const double th = sact[t][0], w0 = sact[t][1], w1 = sact[t][2], w2 = sact[t][3];
            if (t == 0) qm::dyn_core<double, double>(p, th, w0, w1, w2, c, sfd[0][0], sfd[0][1], sfd[0][2]);
            else if (FT) qm::dyn_core<double, double, false>(p, th, w0, w1, w2, c, sfd[t][0], sfd[t][1], sfd[t][2]);
            else qm::dyn_core<double, double, false>(p, th, w0, w1, w2, c, 0.0, 0.0, 0.0);
        }
    }
    if (lane == 0) adj_store_state(p, ws + WS_X + 16 * k);
#endif
    // the step with ONE first-order seed per lane (17 lanes): reward gradient and column `lane` of df/dz
    qm::State<qm::D1> s;
#define OP(m, i) s.m = qm::D1{p.m, lane == i ? 1.0 : 0.0};
    ADJ_FOR_STATE(OP)
#undef OP
#if ADJ_FS
    qm::D1 fd[3], fdn[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        fd[i] = qm::D1{pf[i], lane == 13 + i ? 1.0 : 0.0};
        fdn[i] = fd[i];
    }
#endif
    qm::D1 r{0.0, 0.0};
    if (k >= 1) {
        double tar[6];
        adj_targets(st, A.pos_traj + b * A.traj_stride, A.vel_traj + b * A.traj_stride, A.T, time0, k, tar);
        r = qm::reward_kind<qm::D1, double>(A.reward, s, tar[0], tar[1], tar[2], tar[3], tar[4], tar[5]);
    }
    if (k <= HH - 2) {
        qm::D1 act[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const qm::D1 x{(double)am(4 * k + d), lane == NX + d ? 1.0 : 0.0};
            act[d] = qm::clip11_(qm::clip11_(x));  // quadrotor.py:223 and :258
        }
#if ADJ_FS
        adj_next_force<qm::D1>(A, D, b, k, s, fd, fdn);  // from the PRE-step state
        qm::dyn_step<qm::D1, double, true, qm::D1>(s, act[0], act[1], act[2], act[3], c, fd[0], fd[1], fd[2]);
#else
        qm::dyn_step<qm::D1, double>(s, act[0], act[1], act[2], act[3], c, sfd[k][0], sfd[k][1], sfd[k][2]);
#endif
    }
    if (lane < NZ) {
        if (lane < NX) ws[WS_GL + 16 * k + lane] = r.a;  // 0 for k = 0 (the reward does not see the force: rows 13..15 are zeros)
        if (k <= HH - 2) {
            double *__restrict__ jf = ws + WS_JF + (size_t)k * NX * NZ;
#define OP(m, i) jf[i * NZ + lane] = s.m.a;
            ADJ_FOR_STATE(OP)
#undef OP
#if ADJ_FS
            jf[13 * NZ + lane] = fdn[0].a; jf[14 * NZ + lane] = fdn[1].a; jf[15 * NZ + lane] = fdn[2].a;
#endif
        }
    }
}

// ---- the lambda-independent 9/10 of KM: one hyper-dual step per pair a <= b of the 17 step inputs -> the mixed second derivative of
// r_k and of each of the 13 components of f_k (14 numbers per pair).  M_k = Hess_z(r_k + lam_{k+1} . f_k) is their combination
// with the costate -- which the chains of KC are still computing: these 32 workgroups ride in KC's launch on otherwise idle
// CUs (5 us in the shadow of the 10 us recursions) and contract as soon as their lam_{k+1} is there (below).
__device__ __forceinline__ void adj_hd_pairs(const AdjArgs &A, int k, int b, int tid)
{
    if (tid >= NPAIR) return;
    const float *__restrict__ st = A.state + (size_t)b * COVO_STATE_FLOATS;
    const float *__restrict__ am = A.a_mean + (size_t)b * NA;
    double *__restrict__ ws = A.ws + (size_t)b * WS_COUNT;
    const int time0 = __float_as_int(st[ST_TIME]);
    int q = tid, a = 0;
    while (q >= NZ - a) { q -= NZ - a; ++a; }
    const int bb = a + q;
    qm::State<double> p;
    adj_load_state(p, ws + WS_X + 16 * k);
    qm::HD r;
    qm::State<qm::HD> s;
#if ADJ_FS
    const double pf[3] = {ws[WS_X + 16 * k + 13], ws[WS_X + 16 * k + 14], ws[WS_X + 16 * k + 15]};
    qm::HD fn[3];
    adj_hd_step(st, am, A, b, time0, k, p, a, bb, r, s, pf, fn);
#else
    adj_hd_step(st, am, A, b, time0, k, p, a, bb, r, s);
#endif
    // the pair's second derivatives: of r_k and of the NX components of f_k
    double h[NH];
    h[0] = r.ab;
#define OP(m, i) h[1 + i] = s.m.ab;
    ADJ_FOR_STATE(OP)
#undef OP
#if ADJ_FS
    h[14] = fn[0].ab; h[15] = fn[1].ab; h[16] = fn[2].ab;
#endif
    // ---- KM: M_k = Hess_z( r_k(x) + lam_{k+1} . f_k(z) ), contracted right here as soon as the costate chain of this launch
    // (workgroup 8, dispatched before this one) has stored lam_{k+1}: row k + 1 of WS_LAM still holding KB's "unset" pattern means
    // not yet.  lam_31 is stored first and the chain runs down at ~0.16 us per step, the hyper-dual step above takes ~5 us: only
    // the first few steps wait at all, and none longer than the chain itself (round 3; the contraction used to be a launch of
    // its own, 3 us, with these 14 numbers per pair going through memory).
    // Forward progress rests on IN-ORDER workgroup dispatch: within batch entry b, workgroup (8, b) -- the costate chain -- is
    // placed on a CU before the pollers (9..40, b) of the same entry are (x runs fastest in the dispatch order of a 2-D grid), so
    // a poller never occupies the slot the chain is waiting for, also when the grid (41 x batch workgroups: covo-offline's 300,
    // the env-batched step) is far from co-resident.  HIP does not promise that order; gfx950's dispatcher keeps it.  The spin is
    // therefore bounded (0.5 s of the wall clock): on a time-out the pair contracts with the NaN pattern -- R, Sigma and L of that
    // call are NaN -- AND the handle's sticky COVO_DEVSTAT_ADJOINT bit is raised, so the next call fails with COVO_E_DEVICE
    // instead of planning on NaNs silently (like the Sigma chain's barriers and the exchange).
    double g = h[0];
    if (k <= HH - 2) {
        const double *__restrict__ Lk = ws + WS_LAM + 16 * (k + 1);
        double lam[NX];
        const long long t0 = wall_clock64();
        for (;;) {
            bool unset = false;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                lam[i] = __hip_atomic_load(Lk + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unset |= (unsigned long long)__double_as_longlong(lam[i]) == ADJ_LAM_UNSET;
            }
            if (!unset) break;
            if (wall_clock64() - t0 > 50000000LL) {
                if (A.status != nullptr) __hip_atomic_fetch_or(A.status, COVO_DEVSTAT_ADJOINT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) g = fma(lam[i], h[1 + i], g);
    }
    double *__restrict__ Mxx = ws + WS_MXX + (size_t)k * 256, *__restrict__ Mxu = ws + WS_MXU + (size_t)k * 64;
    double *__restrict__ Muu = ws + WS_MUU + (size_t)k * 16;
    if (bb < NX) {
        Mxx[a * 16 + bb] = g;
        Mxx[bb * 16 + a] = g;
    } else if (a < NX) {
        Mxu[a * 4 + (bb - NX)] = g;
    } else {
        Muu[(a - NX) * 4 + (bb - NX)] = g;
        Muu[(bb - NX) * 4 + (a - NX)] = g;
    }
}

// zero padding of the x block of M_k (rows / columns NX..15): by the hyper-dual workgroup of step k, all of its threads
__device__ __forceinline__ void adj_hd_padding(const AdjArgs &A, int k, int b, int tid, int nthreads)
{
    double *__restrict__ ws = A.ws + (size_t)b * WS_COUNT;
    double *__restrict__ Mxx = ws + WS_MXX + (size_t)k * 256, *__restrict__ Mxu = ws + WS_MXU + (size_t)k * 64;
    for (int e = tid; e < 256; e += nthreads)
        if ((e >> 4) >= NX || (e & 15) >= NX) Mxx[e] = 0.0;
    if (tid < (16 - NX) * 4) Mxu[NX * 4 + tid] = 0.0;
}

// ---- KC: the two linear recursions on the matrix cores.
// S_{k+1} = A_k S_k + B_k E_k is a (13x13).(13x128) product per step: wave w keeps a 16-column tile of S in
// the MFMA C/D layout (lane (lo, hi), register g = row 4g + hi, column lo) -- which is exactly the B operand
// of k-group g of the next step's v_mfma_f64_16x16x4_f64, so the tile never leaves its registers; A_k
// (zero padded to 16x16) comes from LDS as the A operand.  The costate lam_k = grad r_k + A_k^T lam_{k+1} is
// the same product with A^T and the vector in column 0 of a tile (wave 8).  ~330 cycles per step (four
// dependent MFMAs + the MFMA -> operand hazard), 31 steps.  (One lane per column with the vector in
// registers: 2000 cycles per step; 16-lane rows with ds_swizzle broadcasts: 1000.)
// batched launches (round 4): the hyper-dual workgroups as a launch of their own -- inside adj_chain_kernel every workgroup reserves the
// chains' 59 KB of LDS (two per CU), which the 32 hyper-dual workgroups of an instance never touch; 32 instances are 1 312 workgroups
__global__ __launch_bounds__(256) void adj_hd_kernel(const AdjArgs A)
{
    adj_hd_padding(A, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, 256);
    adj_hd_pairs(A, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x);
}

__global__ __launch_bounds__(256) void adj_chain_kernel(const AdjArgs A)
{
    __shared__ double sjf[(HH - 1) * NX * NZ + 1];  // + one zero: what the lanes outside the 13 x 13 block read
    __shared__ double sgl[HH * 16];
    // one chain per WORKGROUP (the f64 MFMA pipe of a SIMD is not shared with another chain: nine chains in one
    // workgroup put 76 steps on one SIMD); the four waves fill LDS together, wave 0 runs the chain
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, lo = lane & 15, hi = lane >> 4;
    const int w = blockIdx.x;  // 0..7: column tile of S, 8: costate, 9..40: the hyper-dual steps of KM (see adj_hd_pairs)
    double *__restrict__ ws = A.ws + (size_t)b * WS_COUNT;
    if (w >= 9) {
        adj_hd_padding(A, w - 9, b, tid, 256);
        adj_hd_pairs(A, w - 9, b, tid);
        return;
    }
    {
        // all global loads in flight before the first LDS store (a rolled loop pays one L2 latency per trip)
        constexpr int NJ = (HH - 1) * NX * NZ, TJ = (NJ + 255) / 256;
        double vj[TJ];
#pragma unroll
        for (int t = 0; t < TJ; ++t) vj[t] = (tid + 256 * t < NJ) ? ws[WS_JF + tid + 256 * t] : 0.0;
        const double vg0 = ws[WS_GL + tid], vg1 = ws[WS_GL + 256 + tid];
#pragma unroll
        for (int t = 0; t < TJ; ++t)
            if (tid + 256 * t < NJ) sjf[tid + 256 * t] = vj[t];
        sgl[tid] = vg0;
        sgl[256 + tid] = vg1;
        if (tid == 0) sjf[NJ] = 0.0;
    }
    __syncthreads();
    if (tid >= 64) return;
    // The chains are fully unrolled (k is a compile-time constant: every LDS / global address is base + immediate) and the lanes
    // outside the 13 x 13 block read a zero slot instead of being masked off: as a rolled loop with `cond ? sjf[..] : 0` operands
    // hipcc spent ~420 of a step's 755 cycles on exec-mask sequences, index arithmetic and accumulator copies around the four
    // dependent MFMAs (KC_PROF stamps; the MFMAs + their result -> operand hazard are ~330).
    constexpr int ZI = (HH - 1) * NX * NZ;
    if (w == 8) {
        // ---- costate: lam_31 = grad r_31, lam_k = grad r_k + A_k^T lam_{k+1}; the vector is column 0 of the tile
        const double *__restrict__ G = sgl;  // grad r_k, staged in LDS: a global load per step would BE the step time
        double *__restrict__ L = ws + WS_LAM;
        f64x4 lam;
#pragma unroll
        for (int r = 0; r < 4; ++r) lam[r] = (lo == 0 && hi + 4 * r < NX) ? G[16 * (HH - 1) + hi + 4 * r] : 0.0;
        // (agent-scope write-through stores: the hyper-dual workgroups of this launch poll these rows, adj_hd_pairs)
        if (lo == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                __hip_atomic_store(&L[16 * (HH - 1) + hi + 4 * r], lam[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&L[16 * HH + hi + 4 * r], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        int offA[4];  // A^T[lo][4g+hi] = df/dz[4g+hi][lo]
        bool okA[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            okA[g] = lo < NX && 4 * g + hi < NX;
            offA[g] = (4 * g + hi) * NZ + lo;
        }
        bool okG[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) okG[r] = lo == 0 && hi + 4 * r < NX;
        double an[4];
        f64x4 gn;
#pragma unroll
        for (int g = 0; g < 4; ++g) an[g] = sjf[okA[g] ? (HH - 2) * NX * NZ + offA[g] : ZI];
#pragma unroll
        for (int r = 0; r < 4; ++r) gn[r] = okG[r] ? G[16 * (HH - 2) + hi + 4 * r] : 0.0;
#pragma unroll
        for (int k = HH - 2; k >= 1; --k) {
            double ac[4];
            f64x4 acc = gn;
#pragma unroll
            for (int g = 0; g < 4; ++g) ac[g] = an[g];
            if (k > 1) {  // next step's operands in flight
#pragma unroll
                for (int g = 0; g < 4; ++g) an[g] = sjf[okA[g] ? (k - 1) * NX * NZ + offA[g] : ZI];
#pragma unroll
                for (int r = 0; r < 4; ++r) gn[r] = okG[r] ? G[16 * (k - 1) + hi + 4 * r] : 0.0;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[g], lam[g], acc, 0, 0, 0);
            lam = acc;
            if (lo == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) __hip_atomic_store(&L[16 * k + hi + 4 * r], lam[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    // ---- sensitivities: tile w = columns 16w .. 16w+15 = the actions of steps 4w .. 4w+3; S_k tile = 0 for k <= 4w
    const int col = 16 * w + lo, tcol = col >> 2, dcol = col & 3, k0 = 4 * w;
    double *__restrict__ S = ws + WS_S;
    for (int k = 0; k <= k0; ++k) {
#pragma unroll
        for (int r = 0; r < 4; ++r) S[((size_t)k * 16 + hi + 4 * r) * NA + col] = 0.0;
    }
    f64x4 sv = {0.0, 0.0, 0.0, 0.0};
    int offA[4];  // A[lo][4g+hi]
    bool okA[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        okA[g] = lo < NX && 4 * g + hi < NX;
        offA[g] = lo * NZ + 4 * g + hi;
    }
    double an[4];
    f64x4 bn;
#pragma unroll
    for (int g = 0; g < 4; ++g) an[g] = sjf[okA[g] ? k0 * NX * NZ + offA[g] : ZI];
#pragma unroll
    for (int r = 0; r < 4; ++r) bn[r] = (tcol == k0 && hi + 4 * r < NX) ? sjf[(k0 * NX + hi + 4 * r) * NZ + NX + dcol] : 0.0;  // B_k E_k
    double *__restrict__ Sp = S + (size_t)hi * NA + col;
#pragma unroll
    for (int k = 0; k < HH - 1; ++k) {
        if (k < k0) continue;  // (uniform)
        double ac[4];
        f64x4 acc = bn;
#pragma unroll
        for (int g = 0; g < 4; ++g) ac[g] = an[g];
        if (k + 1 < HH - 1) {  // next step's operands in flight
#pragma unroll
            for (int g = 0; g < 4; ++g) an[g] = sjf[okA[g] ? (k + 1) * NX * NZ + offA[g] : ZI];
            // B_k E_k is non-zero only while the tile's own four steps enter (uniform branch)
            if (k + 1 <= k0 + 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    bn[r] = (tcol == k + 1 && hi + 4 * r < NX) ? sjf[((k + 1) * NX + hi + 4 * r) * NZ + NX + dcol] : 0.0;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) bn[r] = 0.0;
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[g], sv[g], acc, 0, 0, 0);
        sv = acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) Sp[((size_t)(k + 1) * 16 + 4 * r) * NA] = sv[r];
    }
}

// ---- KD: R = -( sum_k S_k^T Mxx_k S_k  +  action blocks ), one lower 16x16 tile per workgroup, k split over 8 waves
__device__ __forceinline__ void adj_tri_tile(int w, int &ti, int &tj)
{
    ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= w) ++ti;
    tj = w - ti * (ti + 1) / 2;
}

__global__ __launch_bounds__(512) void adj_gemm_kernel(const AdjArgs A)
{
    __shared__ double red[8][4][64];
    __shared__ double sMu[4][64];  // Mxu of the four step indices 4I .. 4I+3 the rows (and, on diagonal tiles, columns) of this tile have
    __shared__ double sMuu[64];    // Muu of the same four steps
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lo = lane & 15, hi = lane >> 4;
    const double *__restrict__ ws = A.ws + (size_t)b * WS_COUNT;
    double *__restrict__ R = A.R + (size_t)b * NA * NA;
    int I, J;
    adj_tri_tile(blockIdx.x, I, J);
    const double *__restrict__ S = ws + WS_S, *__restrict__ Mxx = ws + WS_MXX;
    // S_k[:, 16 I ..] is zero for k <= 4 I (I >= J): k = 4I+1 .. 31, round-robin over the 8 waves.
    // MFMA f64 16x16x4: A[m = lo][kk = hi], B[kk = hi][n = lo]; C/D: column lo, rows hi + 4 r.
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#ifdef KD_PROF
    const long long kp0 = clock64();
#endif
    // A wave keeps at most 63 loads in flight (vmcnt); one more and the launch pays a second memory round trip to data the
    // previous launch has just written from other XCDs (round 1's order -- 26 epilogue loads, then 48 operands, in every
    // wave -- took 8.6 us for ~1 us of MFMA).  So: the <= 48 MFMA operands first, then the 13 sensitivities of the epilogue
    // (finishing waves only); the action blocks Mxu / Muu travel through LDS, one load per thread.
    // at most 4 values of k per wave
    // The contraction index of the FIRST product runs as kk = 4 hi + g (any order serves, A and B agree): a lane then owns
    // 32 contiguous bytes of the row-major Mxx_k and a wave reads each of its 16 lines once (as Mxx[lo][4g + hi] every one of the
    // four loads touched all 16 lines for 8 bytes each -- 2 KiB through the L1 per instruction, and the L1's 64 B/clk is what
    // this kernel waits for: 8 waves x ~55 KiB).  The second product contracts over the C/D row order 4g + hi of Q.
    double ma[4][4], sj[4][4], si[4][4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int k = 4 * I + 1 + wv + 8 * it;
        const bool on = k < HH;
        const double *__restrict__ Sk = S + (size_t)(on ? k : 0) * 16 * NA, *__restrict__ Mk = Mxx + (size_t)(on ? k : 0) * 256;
        const double2 *__restrict__ Mk2 = reinterpret_cast<const double2 *>(Mk + lo * 16 + 4 * hi);
        const double2 m01 = on ? Mk2[0] : make_double2(0.0, 0.0), m23 = on ? Mk2[1] : make_double2(0.0, 0.0);
        ma[it][0] = m01.x;  // Mxx[m = lo][4 hi + g]
        ma[it][1] = m01.y;
        ma[it][2] = m23.x;
        ma[it][3] = m23.y;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            sj[it][g] = on ? Sk[(size_t)(4 * hi + g) * NA + 16 * J + lo] : 0.0;  // S[4 hi + g][n = 16J + lo]
            si[it][g] = on ? Sk[(size_t)(4 * g + hi) * NA + 16 * I + lo] : 0.0;  // S^T: A[m = lo][kk = hi] = S[4g + hi][16I + lo]
        }
    }
    // epilogue operands of the element this lane will finish (waves 0..3: register wv of the tile)
    const int i = 16 * I + hi + 4 * (wv & 3), j = 16 * J + lo;
    const int ti = i >> 2, di = i & 3, tj = j >> 2, dj = j & 3;
    const int tk = ti > tj ? ti : tj, dk = ti > tj ? di : dj;
    double es[NX];
    if (wv < 4) {
        const int col = ti > tj ? j : i;
        const double *__restrict__ Sk = S + (size_t)tk * 16 * NA;
#pragma unroll
        for (int m = 0; m < NX; ++m) es[m] = Sk[(size_t)m * NA + col];
    } else if (wv == 4) {
        sMuu[lane] = ws[WS_MUU + (size_t)(4 * I) * 16 + lane];
    }
    if (tid < 256) sMu[wv][lane] = ws[WS_MXU + (size_t)(4 * I + wv) * 64 + lane];  // [step][m * 4 + d], m < 16 (rows 13..15 zero)
#ifdef KD_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long kp1 = clock64();
#endif
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (4 * I + 1 + wv + 8 * it >= HH) break;
        f64x4 Q = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < 4; ++g) Q = __builtin_amdgcn_mfma_f64_16x16x4f64(ma[it][g], sj[it][g], Q, 0, 0, 0);
        // Q = Mxx S_J in C layout: register g of lane (lo, hi) is row 4g + hi -- the B operand of k-group g as is
#pragma unroll
        for (int g = 0; g < 4; ++g) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(si[it][g], Q[g], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv][r][lane] = acc[r];
#ifdef KD_PROF
    const long long kp2 = clock64();
#endif
    __syncthreads();
#ifdef KD_PROF
    const long long kp3 = clock64();
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 35)) printf("KD wg %d: loads %lld mfma %lld barrier %lld\n", (int)blockIdx.x, kp1 - kp0, kp2 - kp1, kp3 - kp2);
#endif
    double v = 0.0;
    if (wv < 4) {
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) v += red[w8][wv][lane];
        // action blocks: u_i enters at step t_i where column j has sensitivity S_{t_i}[:, j] (t_j < t_i), and vice versa
        if (ti != tj) {
            const double *mu = sMu[tk - 4 * I];
#pragma unroll
            for (int m = 0; m < NX; ++m) v = fma(es[m], mu[m * 4 + dk], v);
        } else {
            v += sMuu[(ti - 4 * I) * 16 + di * 4 + dj];
        }
        // C = -J.  Off-diagonal tiles mirror; in diagonal tiles the lower half writes both copies (exactly symmetric R)
        if (I != J || i >= j) {
            R[(size_t)i * NA + j] = -v;
            R[(size_t)j * NA + i] = -v;
        }
    }
    if (A.stats.rpart != nullptr) {
        // the Sigma chain's input statistics of R = -v, tile by tile (sym_stats.hpp): the chain then needs no prep launch.  In a
        // diagonal tile R holds the lower half and its mirror image, so the statistics take the mirrored values too
        __shared__ double st_tmp[16][17];
        __shared__ double st_part[4];
        __shared__ double st_diag[16][17];
        double rv = -v;
        if (I == J) {
            if (wv < 4) st_diag[hi + 4 * wv][lo] = rv;
            __syncthreads();
            if (wv < 4 && i < j) rv = st_diag[lo][hi + 4 * wv];
        }
        SymStatsOut so = A.stats;  // this instance's block
        so.rpart += (size_t)b * so.stride;
        so.fpart += (size_t)b * so.stride;
        so.diag += (size_t)b * so.stride;
        sym_tile_stats(wv < 4, rv, I, J, (int)blockIdx.x, lane, wv & 3, so, st_tmp, st_part);
        // the chain's first launch may be the persistent one: its barrier flags are cleared here, a launch ahead
        if (so.flags != nullptr && blockIdx.x == 0 && tid < 64) so.flags[(size_t)b * so.stride + tid] = 0.0;
        if (so.ready != nullptr && blockIdx.x == 0 && tid == 64) so.ready[(size_t)b * so.stride] = 0.0;
    }
}
#undef ADJ_FOR_STATE
