/* Body of the plain-C oracle, included twice by covo_oracle.c with
 *   REAL = float  / SUF(x) = x##_f32   and   REAL = double / SUF(x) = x##_f64.
 * TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/__init__.py).
 * Reference lines are relative to /root/reference/quadjax.
 *
 * Geometry is in closed form: for unit q=(x,y,z,w)
 *   qtoQ(q)[:,2] = [2(xz+yw), 2(yz-xw), w^2-x^2-y^2+z^2]      (geom.py:68-77)
 *   0.5 L(q) H omega = 0.5 [w*om + v x om, -v.om]             (geom.py:41-55, free.py:96)
 * tests/test_oracle.py checks these against the literal 4x4-matmul form of ref_np.py.
 */

typedef struct {
    REAL pos[3], vel[3], quat[4], omega[3], f[3], pos_tar[3], vel_tar[3];
    int time;
} SUF(ostate);

static inline REAL SUF(norm3)(const REAL *v) { return SQRT(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* dynamics/utils.py:266-274 */
static inline REAL SUF(clip01)(REAL x) { return x < (REAL)0 ? (REAL)0 : (x > (REAL)1 ? (REAL)1 : x); }
static inline REAL SUF(log_pos)(REAL e)
{
    REAL l = LOG(e + (REAL)1);
    return e * (REAL)0.4 + SUF(clip01)(l * 4) * (REAL)0.4 + SUF(clip01)(l * 8) * (REAL)0.2 +
           SUF(clip01)(l * 16) * (REAL)0.1 + SUF(clip01)(l * 32) * (REAL)0.1;
}

/* dynamics/utils.py:285-294 -- yaw from the STORED (possibly un-normalised) quaternion */
static inline REAL SUF(reward)(const SUF(ostate) * s)
{
    REAL dp[3], dv[3];
    for (int i = 0; i < 3; ++i) { dp[i] = s->pos_tar[i] - s->pos[i]; dv[i] = s->vel_tar[i] - s->vel[i]; }
    REAL err_pos = SUF(norm3)(dp), err_vel = SUF(norm3)(dv);
    const REAL *q = s->quat;
    REAL yaw = ATAN2(2 * (q[3] * q[2] + q[0] * q[1]), 1 - 2 * (q[1] * q[1] + q[2] * q[2]));
    return (REAL)1.3 - (REAL)0.05 * err_vel - SUF(log_pos)(err_pos) - FABS(yaw) * (REAL)0.2;
}

/* dynamics/utils.py:297-313 (task "tracking_slow", envs/quadrotor.py:58-66) */
static inline REAL SUF(reward_realworld)(const SUF(ostate) * s)
{
    REAL pe = 0;
    for (int i = 0; i < 3; ++i) { REAL d = s->pos[i] - s->pos_tar[i]; pe = pe + d * d; }
    REAL pos_err = pe / (REAL)3;                      /* jnp.mean(pos_err**2) */
    REAL quat_err = (REAL)1 - s->quat[3] * s->quat[3];
    REAL cost = (REAL)5.0 * pos_err + (REAL)3.0 * quat_err;
    cost = cost * (REAL)0.02;
    return -cost;
}
static inline REAL SUF(reward_kind)(const SUF(ostate) * s, int kind) { return kind == 1 ? SUF(reward_realworld)(s) : SUF(reward)(s); }

/* dynamics/free.py:10-58: the disturbance the NEXT step uses, from the PRE-step state (free.py:147).
 * dist = [kind (2 periodic, 3 sin, 4 drag, 5 mixed), disturb_period, disturb_scale, disturb_params[6]];
 * draw[3] = the explicit value of uniform(disturb_key, (3,), -scale, scale) for this call (periodic / mixed). */
static inline void SUF(disturb_sin)(const SUF(ostate) * s, const double *dist, REAL *out)
{
    const REAL two_pi = (REAL)2 * (REAL)3.14159265358979323846;
    for (int i = 0; i < 3; ++i) {
        REAL scale = (REAL)dist[3 + i] * (REAL)dist[2];
        REAL period = (REAL)dist[3 + i] * (REAL)(dist[1] / 3) + (REAL)dist[1];
        REAL phase = (REAL)dist[6 + i] * (REAL)2 * (REAL)3.14159265358979323846;
        out[i] = scale * SIN(two_pi / period * (REAL)s->time + phase);
    }
}
static inline void SUF(disturb_drag)(const SUF(ostate) * s, const double *dist, REAL *out)
{
    for (int i = 0; i < 3; ++i) {
        REAL rel = s->vel[i] - (REAL)dist[3 + i] * (REAL)0.5;
        out[i] = -FABS((REAL)dist[2]) * rel * FABS(rel) / (REAL)(1.5 * 1.5);
    }
}
static inline void SUF(disturb_period)(const SUF(ostate) * s, const double *dist, const REAL *draw, REAL *out)
{
    const int hit = (s->time % (int)dist[1]) == 0;
    for (int i = 0; i < 3; ++i) out[i] = hit ? draw[i] : s->f[i];
}
static inline void SUF(disturb_next)(const SUF(ostate) * s, const double *dist, const REAL *draw, REAL *out)
{
    const int kind = (int)dist[0];
    if (kind == 2) SUF(disturb_period)(s, dist, draw, out);
    else if (kind == 3) SUF(disturb_sin)(s, dist, out);
    else if (kind == 4) SUF(disturb_drag)(s, dist, out);
    else if (kind == 5) {
        REAL a[3], b[3], c[3];
        SUF(disturb_drag)(s, dist, a);
        SUF(disturb_sin)(s, dist, b);
        SUF(disturb_period)(s, dist, draw, c);
        for (int i = 0; i < 3; ++i) out[i] = (a[i] + b[i] + c[i]) / (REAL)3;
    } else {
        out[0] = out[1] = out[2] = 0;
    }
}

/* envs/quadrotor.py:479-490; rollover = !disable_rollover_terminate (:486-490) */
static inline int SUF(terminal)(const SUF(ostate) * s, int max_steps, int rollover)
{
    int done = (s->time >= max_steps) || FABS(s->pos[0]) > 3 || FABS(s->pos[1]) > 3 || FABS(s->pos[2]) > 3;
    if (rollover)
        done = done || s->quat[3] < (REAL)0.70710678118654752440 /* cos(pi/4) */ || FABS(s->omega[0]) > 100 ||
               FABS(s->omega[1]) > 100 || FABS(s->omega[2]) > 100;
    return done;
}

/* envs/quadrotor.py:250-263 + dynamics/free.py:114-155 + free.py:74-112.
 * prm = [max_thrust, max_torque(3), max_omega(3), dt, g, m, action_scale, alpha_bodyrate] */
static inline void SUF(dyn_step)(SUF(ostate) * s, const REAL *act, const double *prm, const REAL *f_next,
                                 const REAL *pos_traj, const REAL *vel_traj, int T)
{
    const REAL max_thrust = (REAL)prm[0], dt = (REAL)prm[7], g = (REAL)prm[8], m = (REAL)prm[9];
    const REAL ascale = (REAL)prm[10], alpha = (REAL)prm[11];
    REAL a[4];
    for (int i = 0; i < 4; ++i) a[i] = act[i] < -1 ? -1 : (act[i] > 1 ? 1 : act[i]); /* quadrotor.py:223,258 */
    REAL thrust = (a[0] + (REAL)1) / (REAL)2 * max_thrust;                           /* :259 */
    REAL u[4];
    u[0] = thrust * ascale; /* free.py:82 */
    for (int i = 0; i < 3; ++i) {
        REAL torque = a[1 + i] * (REAL)prm[1 + i];                          /* quadrotor.py:260 */
        REAL omega_tar = torque / (REAL)prm[1 + i] * (REAL)prm[4 + i];      /* free.py:122 */
        u[1 + i] = omega_tar * ascale;                                      /* free.py:82 */
    }
    REAL qn = SQRT(s->quat[0] * s->quat[0] + s->quat[1] * s->quat[1] + s->quat[2] * s->quat[2] +
                   s->quat[3] * s->quat[3]);
    REAL x = s->quat[0] / qn, y = s->quat[1] / qn, z = s->quat[2] / qn, w = s->quat[3] / qn; /* free.py:88 */
    const REAL *om = s->omega;
    REAL Qz[3] = {2 * (x * z + y * w), 2 * (y * z - x * w), w * w - x * x - y * y + z * z};
    REAL qd[4] = {(REAL)0.5 * (w * om[0] + (y * om[2] - z * om[1])), (REAL)0.5 * (w * om[1] + (z * om[0] - x * om[2])),
                  (REAL)0.5 * (w * om[2] + (x * om[1] - y * om[0])), (REAL)0.5 * -(x * om[0] + y * om[1] + z * om[2])};
    REAL vd[3];
    for (int i = 0; i < 3; ++i) vd[i] = (i == 2 ? -g : (REAL)0) + (REAL)1 / m * (Qz[i] * u[0] + s->f[i]); /* :97-99 */
    REAL qq[4] = {x + qd[0] * dt, y + qd[1] * dt, z + qd[2] * dt, w + qd[3] * dt};                        /* :103 */
    for (int i = 0; i < 3; ++i) {
        s->pos[i] = s->pos[i] + s->vel[i] * dt;                            /* :102 (old v) */
        s->vel[i] = s->vel[i] + vd[i] * dt;                                /* :104 */
        s->omega[i] = alpha * om[i] + ((REAL)1 - alpha) * u[1 + i];        /* :105-107 */
    }
    REAL qn2 = SQRT(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
    for (int i = 0; i < 4; ++i) s->quat[i] = qq[i] / qn2;                  /* free.py:139 */
    for (int i = 0; i < 3; ++i) s->f[i] = f_next[i];                       /* free.py:147 */
    s->time += 1;                                                          /* :150 */
    int idx = s->time < 0 ? 0 : (s->time > T - 1 ? T - 1 : s->time);       /* gather clamps */
    for (int i = 0; i < 3; ++i) { s->pos_tar[i] = pos_traj[3 * idx + i]; s->vel_tar[i] = vel_traj[3 * idx + i]; }
}

/* controllers/covo.py:227-263 / mppi.py:71-106.
 * state22 = [pos vel quat omega f_disturb pos_tar vel_tar]; a is (N,H,4) row-major.
 * rewards (N,H) and poses (H,N,3) may be NULL. */
void SUF(oracle_rollout_ex)(const double *prm, int max_steps, const REAL *state22, int time, const REAL *pos_traj,
                            const REAL *vel_traj, int T, const REAL *a, long N, int H, REAL discount,
                            const REAL *f_shared, REAL *cost, REAL *rewards, REAL *poses, int rollover, int reward_kind,
                            const double *dist)
{
#pragma omp parallel for schedule(static)
    for (long n = 0; n < N; ++n) {
        SUF(ostate) s;
        memcpy(s.pos, state22, 3 * sizeof(REAL));
        memcpy(s.vel, state22 + 3, 3 * sizeof(REAL));
        memcpy(s.quat, state22 + 6, 4 * sizeof(REAL));
        memcpy(s.omega, state22 + 10, 3 * sizeof(REAL));
        memcpy(s.f, state22 + 13, 3 * sizeof(REAL));
        memcpy(s.pos_tar, state22 + 16, 3 * sizeof(REAL));
        memcpy(s.vel_tar, state22 + 19, 3 * sizeof(REAL));
        s.time = time;
        REAL reward_before = 0, acc = 0, disc = 1;
        int done_before = 0;
        for (int k = 0; k < H; ++k) {
            REAL r = SUF(reward_kind)(&s, reward_kind); /* quadrotor.py:243 (pre-step) */
            int done = SUF(terminal)(&s, max_steps, rollover); /* quadrotor.py:244 */
            REAL fn[3] = {f_shared[0], f_shared[1], f_shared[2]};
            /* dist != NULL: a state/time-dependent model (free.py:10-58) evaluated on the pre-step state; f_shared is
             * then the ONE draw all steps share (every step_env call of the scan has the same key, covo.py:225,231) */
            if (dist) SUF(disturb_next)(&s, dist, f_shared, fn);
            SUF(dyn_step)(&s, a + ((size_t)n * H + k) * 4, prm, fn, pos_traj, vel_traj, T);
            if (done_before) r = reward_before; /* covo.py:233 */
            done_before |= done;
            reward_before = r;
            acc += r * disc; /* covo.py:257-261 */
            disc *= discount;
            if (rewards) rewards[(size_t)n * H + k] = r;
            if (poses) memcpy(poses + ((size_t)k * N + n) * 3, s.pos, 3 * sizeof(REAL));
        }
        cost[n] = -acc; /* covo.py:263 */
    }
}

void SUF(oracle_rollout)(const double *prm, int max_steps, const REAL *state22, int time, const REAL *pos_traj,
                         const REAL *vel_traj, int T, const REAL *a, long N, int H, REAL discount,
                         const REAL *f_shared, REAL *cost, REAL *rewards, REAL *poses, int rollover)
{
    SUF(oracle_rollout_ex)(prm, max_steps, state22, time, pos_traj, vel_traj, T, a, N, H, discount, f_shared, cost, rewards,
                           poses, rollover, 0, NULL);
}

/* controllers/covo.py:266-275 as an online-softmax partial over one shard:
 * m = min cost, s = sum exp(-(c-m)/lam), v[n] = sum exp(..) * a[.,n].  a is (N,n) row-major. */
void SUF(oracle_softmax_partial)(const REAL *cost, const REAL *a, long N, int n, REAL lam, REAL *m_out, REAL *s_out,
                                 REAL *v_out)
{
    REAL m = cost[0];
    for (long i = 1; i < N; ++i) m = cost[i] < m ? cost[i] : m;
    double s = 0;
    double *v = (double *)calloc((size_t)n, sizeof(double));
    for (long i = 0; i < N; ++i) {
        REAL w = EXP(-(cost[i] - m) / lam);
        if (w == 0) continue;
        s += w;
        for (int j = 0; j < n; ++j) v[j] += (double)w * (double)a[(size_t)i * n + j];
    }
    *m_out = m;
    *s_out = (REAL)s;
    for (int j = 0; j < n; ++j) v_out[j] = (REAL)v[j];
    free(v);
}
