// quad_model.hpp -- one-step quadrotor model shared by the fp32 rollout kernel and the
// fp64 hyper-dual Hessian kernel (gfx950 device code).
//
// Restates, in closed form, what XLA lowers from (paths relative to /root/reference/quadjax):
//   envs/quadrotor.py:215-263   step_env / raw_step (clip, thrust & body-rate maps)
//   dynamics/free.py:74-155     quad_dynamics_bodyrate + free_dynamics_3d_bodyrate
//   dynamics/geom.py:41-77      L(q), H, qtoQ   (only Q[:,2] and 0.5 L(q) H w are needed)
//   dynamics/utils.py:266-294   log_pos_fn, tracking_penyaw_reward_fn
//   envs/quadrotor.py:479-490   is_terminal (rollover check disabled by main(), :779)
//
// The scalar type S is `float` (rollout: hardware v_rsq/v_sqrt/v_log/v_rcp, 1 ulp each) or
// HyperDual (Hessian: value + two first-order parts + one mixed second-order part, fp64).
// U is the type of wave-uniform quantities (float / double).
#pragma once
#include <hip/hip_runtime.h>

namespace qm {

// ---------------------------------------------------------------- float primitives
__device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float rsqrt_(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float log_(float x) { return __builtin_amdgcn_logf(x) * 0.69314718056f; }
__device__ __forceinline__ float abs_(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ float sat01_(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
__device__ __forceinline__ float clip11_(float x) { return __builtin_amdgcn_fmed3f(x, -1.0f, 1.0f); }
// jnp.clip's NaN semantics (minimum(maximum(x, lo), hi) PROPAGATES a NaN; v_med3 / v_max / v_min return the other operand):
// COVO_FLAG_PROPAGATE_NAN -- v_cmp_u + v_cndmask on top of the v_med3
__device__ __forceinline__ float clip11_nan_(float x) { return (x != x) ? x : __builtin_amdgcn_fmed3f(x, -1.0f, 1.0f); }
__device__ __forceinline__ float value_(float x) { return x; }

// |atan2(y, x)|: odd minimax polynomial of atan on [0,1] (max abs error 1.2e-7) + octant fix-up.
__device__ __forceinline__ float atan2abs_(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(mx);
    const float s = t * t;
    float r = 0.00282363896f;
    r = __builtin_fmaf(r, s, -0.0159569028f);
    r = __builtin_fmaf(r, s, 0.0425049886f);
    r = __builtin_fmaf(r, s, -0.0748900920f);
    r = __builtin_fmaf(r, s, 0.106347933f);
    r = __builtin_fmaf(r, s, -0.142027363f);
    r = __builtin_fmaf(r, s, 0.199926957f);
    r = __builtin_fmaf(r, s, -0.333331018f);
    r = r * s;
    r = __builtin_fmaf(r, t, t);
    r = (mx == 0.0f) ? 0.0f : r;              // atan2(0,0) = 0
    r = (ay > ax) ? (1.57079637f - r) : r;
    r = (x < 0.0f) ? (3.14159274f - r) : r;
    return r;
}

// ---------------------------------------------------------------- double primitives (primal prefix of the Hessian)
// 1/sqrt(x) and 1/x without the libm division/sqrt expansions (~45-60 instructions each, and a lone fp64
// wave issues one instruction per ~5 cycles): hardware seed (v_rsq_f64 / v_rcp_f64, ~2^-24 relative,
// scripts/probe) + two Newton steps -> ~2e-16.  x = 0 gives NaN (libm: inf); every use below either has
// x > 0 (quaternion norms, 1 + err_pos, yaw denominator) or propagates NaN tangents at 0 exactly like
// jnp.linalg.norm's JVP does.
__device__ __forceinline__ double rsq64_(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-(x * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-(x * y), y, 1.0);
    return fma(0.5 * y, e, y);
}
__device__ __forceinline__ double rcp64_(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
__device__ __forceinline__ double sqrt_(double x) { return (x == 0.0) ? 0.0 : x * rsq64_(x); }
__device__ __forceinline__ double rsqrt_(double x) { return rsq64_(x); }
__device__ __forceinline__ double log_(double x) { return log(x); }
__device__ __forceinline__ double abs_(double x) { return fabs(x); }
__device__ __forceinline__ double sat01_(double x) { return fmin(fmax(x, 0.0), 1.0); }
__device__ __forceinline__ double clip11_(double x) { return fmin(fmax(x, -1.0), 1.0); }
__device__ __forceinline__ double value_(double x) { return x; }
__device__ __forceinline__ double atan2abs_(double y, double x) { return fabs(atan2(y, x)); }

// ---------------------------------------------------------------- hyper-dual fp64
// x = v + a e1 + b e2 + ab e1 e2  (e1^2 = e2^2 = 0).  g(x) = g(v) + g'(v)(a e1 + b e2)
//   + (g'(v) ab + g''(v) a b) e1 e2  -- forward-over-forward AD of jax.jacfwd(jax.jacfwd(.)).
struct HD {
    double v, a, b, ab;
};
__device__ __forceinline__ HD hd(double v) { return HD{v, 0.0, 0.0, 0.0}; }
__device__ __forceinline__ HD operator+(HD x, HD y) { return HD{x.v + y.v, x.a + y.a, x.b + y.b, x.ab + y.ab}; }
__device__ __forceinline__ HD operator-(HD x, HD y) { return HD{x.v - y.v, x.a - y.a, x.b - y.b, x.ab - y.ab}; }
__device__ __forceinline__ HD operator-(HD x) { return HD{-x.v, -x.a, -x.b, -x.ab}; }
__device__ __forceinline__ HD operator+(HD x, double c) { return HD{x.v + c, x.a, x.b, x.ab}; }
__device__ __forceinline__ HD operator+(double c, HD x) { return HD{x.v + c, x.a, x.b, x.ab}; }
__device__ __forceinline__ HD operator-(HD x, double c) { return HD{x.v - c, x.a, x.b, x.ab}; }
__device__ __forceinline__ HD operator-(double c, HD x) { return HD{c - x.v, -x.a, -x.b, -x.ab}; }
__device__ __forceinline__ HD operator*(HD x, double c) { return HD{x.v * c, x.a * c, x.b * c, x.ab * c}; }
__device__ __forceinline__ HD operator*(double c, HD x) { return HD{x.v * c, x.a * c, x.b * c, x.ab * c}; }
__device__ __forceinline__ HD operator*(HD x, HD y)
{
    return HD{x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b,
              x.ab * y.v + x.a * y.b + x.b * y.a + x.v * y.ab};
}
// unary chain rule with f = g(v), d1 = g'(v), d2 = g''(v)
__device__ __forceinline__ HD chain_(HD x, double f, double d1, double d2)
{
    return HD{f, d1 * x.a, d1 * x.b, d1 * x.ab + d2 * x.a * x.b};
}
__device__ __forceinline__ HD sqrt_(HD x)
{
    const double r = rsq64_(x.v);            // NaN at 0 => NaN tangents, as jnp.linalg.norm's JVP at 0
    const double s = (x.v == 0.0) ? 0.0 : x.v * r;
    return chain_(x, s, 0.5 * r, -0.25 * r * r * r);
}
__device__ __forceinline__ HD rsqrt_(HD x)
{
    const double r = rsq64_(x.v), r2 = r * r, r3 = r2 * r;
    return chain_(x, r, -0.5 * r3, 0.75 * r3 * r2);
}
__device__ __forceinline__ HD log_(HD x)
{
    const double r = rcp64_(x.v);
    return chain_(x, log(x.v), r, -r * r);
}
// lax.abs JVP: g * sign(x), sign(0) = 0
__device__ __forceinline__ HD abs_(HD x)
{
    const double sg = (x.v > 0.0) ? 1.0 : ((x.v < 0.0) ? -1.0 : 0.0);
    return HD{fabs(x.v), sg * x.a, sg * x.b, sg * x.ab};
}
// jnp.clip = minimum(maximum(x, lo), hi); lax.max/min JVPs split a tie 0.5/0.5 (_balanced_eq)
__device__ __forceinline__ HD clip_(HD x, double lo, double hi)
{
    double g = 1.0;
    double v = x.v;
    if (v < lo) { v = lo; g = 0.0; } else if (v == lo) { g = 0.5; }
    if (v > hi) { v = hi; g = 0.0; } else if (v == hi) { g *= 0.5; }
    return HD{v, g * x.a, g * x.b, g * x.ab};
}
__device__ __forceinline__ HD sat01_(HD x) { return clip_(x, 0.0, 1.0); }
__device__ __forceinline__ HD clip11_(HD x) { return clip_(x, -1.0, 1.0); }
__device__ __forceinline__ double value_(HD x) { return x.v; }
// |atan2(y, x)| = abs_(atan2(y, x))
__device__ __forceinline__ HD atan2abs_(HD y, HD x)
{
    const double inv = rcp64_(x.v * x.v + y.v * y.v), inv2 = inv * inv;
    const double f = atan2(y.v, x.v);
    const double fy = x.v * inv, fx = -y.v * inv;  // d/dy, d/dx
    const double fyy = -2.0 * x.v * y.v * inv2;
    const double fxx = -fyy;
    const double fxy = (y.v * y.v - x.v * x.v) * inv2;
    HD r;
    r.v = f;
    r.a = fy * y.a + fx * x.a;
    r.b = fy * y.b + fx * x.b;
    r.ab = fy * y.ab + fx * x.ab + fyy * y.a * y.b + fxx * x.a * x.b + fxy * (y.a * x.b + x.a * y.b);
    return abs_(r);
}

// ---------------------------------------------------------------- first-order dual fp64
// x = v + a e (e^2 = 0): one forward-mode tangent.  Used for the step Jacobians of the adjoint Hessian
// (hessian_adj.hip: 17 seeds per step); same conventions as HD above (clip ties 0.5, |x|' = sign, norm'(0) = NaN).
struct D1 {
    double v, a;
};
__device__ __forceinline__ D1 operator+(D1 x, D1 y) { return D1{x.v + y.v, x.a + y.a}; }
__device__ __forceinline__ D1 operator-(D1 x, D1 y) { return D1{x.v - y.v, x.a - y.a}; }
__device__ __forceinline__ D1 operator-(D1 x) { return D1{-x.v, -x.a}; }
__device__ __forceinline__ D1 operator+(D1 x, double c) { return D1{x.v + c, x.a}; }
__device__ __forceinline__ D1 operator+(double c, D1 x) { return D1{x.v + c, x.a}; }
__device__ __forceinline__ D1 operator-(D1 x, double c) { return D1{x.v - c, x.a}; }
__device__ __forceinline__ D1 operator-(double c, D1 x) { return D1{c - x.v, -x.a}; }
__device__ __forceinline__ D1 operator*(D1 x, double c) { return D1{x.v * c, x.a * c}; }
__device__ __forceinline__ D1 operator*(double c, D1 x) { return D1{x.v * c, x.a * c}; }
__device__ __forceinline__ D1 operator*(D1 x, D1 y) { return D1{x.v * y.v, x.a * y.v + x.v * y.a}; }
__device__ __forceinline__ D1 sqrt_(D1 x)
{
    const double r = rsq64_(x.v);
    return D1{(x.v == 0.0) ? 0.0 : x.v * r, 0.5 * r * x.a};
}
__device__ __forceinline__ D1 rsqrt_(D1 x)
{
    const double r = rsq64_(x.v);
    return D1{r, -0.5 * r * r * r * x.a};
}
__device__ __forceinline__ D1 log_(D1 x) { return D1{log(x.v), rcp64_(x.v) * x.a}; }
__device__ __forceinline__ D1 abs_(D1 x)
{
    const double sg = (x.v > 0.0) ? 1.0 : ((x.v < 0.0) ? -1.0 : 0.0);
    return D1{fabs(x.v), sg * x.a};
}
__device__ __forceinline__ D1 clip_(D1 x, double lo, double hi)
{
    double g = 1.0;
    double v = x.v;
    if (v < lo) { v = lo; g = 0.0; } else if (v == lo) { g = 0.5; }
    if (v > hi) { v = hi; g = 0.0; } else if (v == hi) { g *= 0.5; }
    return D1{v, g * x.a};
}
__device__ __forceinline__ D1 sat01_(D1 x) { return clip_(x, 0.0, 1.0); }
__device__ __forceinline__ D1 clip11_(D1 x) { return clip_(x, -1.0, 1.0); }
__device__ __forceinline__ double value_(D1 x) { return x.v; }
__device__ __forceinline__ D1 atan2abs_(D1 y, D1 x)
{
    const double inv = rcp64_(x.v * x.v + y.v * y.v);
    return abs_(D1{atan2(y.v, x.v), (x.v * y.a - y.v * x.a) * inv});
}

// ---------------------------------------------------------------- model
template <class S>
struct State {
    S px, py, pz, vx, vy, vz, qx, qy, qz, qw, ox, oy, oz;
};

// wave-uniform constants derived once from covo_env_params
template <class U>
struct Consts {
    U thrust_half;  // 0.5 * max_thrust * action_scale :  u0 = (a0 + 1) * thrust_half
    U komega[3];    // max_omega * action_scale        :  omega_tar = a[1:] * komega
    U dt, half_dt, neg_g, inv_m, alpha, one_m_alpha, pos_limit;
};

// dynamics/utils.py:289-290: yaw = atan2(yn, yd) of the STORED quaternion
template <class S, class U>
__device__ __forceinline__ void yaw_terms(const State<S> &s, S &yn, S &yd)
{
    yn = U(2) * (s.qw * s.qz + s.qx * s.qy);
    yd = U(1) - U(2) * (s.qy * s.qy + s.qz * s.qz);
}

// dynamics/utils.py:285-294 from position, velocity and the yaw terms
template <class S, class U>
__device__ __forceinline__ S reward_parts(S px, S py, S pz, S vx, S vy, S vz, S yn, S yd, U tx, U ty, U tz, U tvx, U tvy,
                                          U tvz)
{
    const S dx = tx - px, dy = ty - py, dz = tz - pz;
    const S ex = tvx - vx, ey = tvy - vy, ez = tvz - vz;
    const S err_pos = sqrt_(dx * dx + dy * dy + dz * dz);
    const S err_vel = sqrt_(ex * ex + ey * ey + ez * ez);
    const S yaw = atan2abs_(yn, yd);
    const S l = log_(err_pos + U(1));
    const S lp = err_pos * U(0.4) + sat01_(l * U(4)) * U(0.4) + sat01_(l * U(8)) * U(0.2) +
                 sat01_(l * U(16)) * U(0.1) + sat01_(l * U(32)) * U(0.1);
    return U(1.3) - err_vel * U(0.05) - lp - yaw * U(0.2);
}

// dynamics/utils.py:285-294 on the PRE-step state (envs/quadrotor.py:243)
template <class S, class U>
__device__ __forceinline__ S reward(const State<S> &s, U tx, U ty, U tz, U tvx, U tvy, U tvz)
{
    S yn, yd;
    yaw_terms<S, U>(s, yn, yd);
    return reward_parts<S, U>(s.px, s.py, s.pz, s.vx, s.vy, s.vz, yn, yd, tx, ty, tz, tvx, tvy, tvz);
}

// dynamics/utils.py:297-313 (tracking_realworld_reward_fn, task "tracking_slow"): -0.02 (5 mean((p - p_tar)^2) + 3 (1 - w^2)) with
// the STORED quaternion's w
template <class S, class U>
__device__ __forceinline__ S reward_realworld(const State<S> &s, U tx, U ty, U tz)
{
    const S dx = s.px - tx, dy = s.py - ty, dz = s.pz - tz;
    const S pos_err = (dx * dx + dy * dy + dz * dz) * (U(1) / U(3));
    const S quat_err = U(1) - s.qw * s.qw;
    return -((pos_err * U(5) + quat_err * U(3)) * U(0.02));
}

// env.reward_fn by covo_env_params.reward_kind (0: penyaw, 1: realworld; wave-uniform)
template <class S, class U>
__device__ __forceinline__ S reward_kind(int kind, const State<S> &s, U tx, U ty, U tz, U tvx, U tvy, U tvz)
{
    if (kind == 1) return reward_realworld<S, U>(s, tx, ty, tz);
    return reward<S, U>(s, tx, ty, tz, tvx, tvy, tvz);
}

// dynamics/free.py:41-47 with the model's constant folded: coeff = c_drag * (-|disturb_scale| / 1.5^2), off = disturb_params[:3] / 2
template <class S, class U>
__device__ __forceinline__ S drag_force(S v, U off, U coeff)
{
    const S rel = v - off;
    return (rel * abs_(rel)) * coeff;
}

// envs/quadrotor.py:250-263 + dynamics/free.py:114-155 (+74-112).  a* are already clipped.
// (fx,fy,fz) = f_disturb acting during THIS step (free.py:91,98).
// ENTRY_NORM = false: the stored quaternion is the one free.py:139 normalised at the end of the previous step -- normalising it
// again on entry (free.py:88) moves it by <= 1 ulp; used only for the plain primal prefix of the adjoint Hessian (steps >= 1),
// never where derivatives are taken.
// dyn_core: the step given the action's thrust and body-rate targets (dyn_step = action terms + dyn_core, same operations)
// F: the type of the disturbance (U when it is wave-uniform, S when it is part of the differentiated / per-sample state)
template <class S, class U, bool ENTRY_NORM = true, class F = U>
__device__ __forceinline__ void dyn_core(State<S> &s, S thrust, S wtx, S wty, S wtz, const Consts<U> &c, F fx, F fy, F fz);

template <class S, class U, bool ENTRY_NORM = true, class F = U>
__device__ __forceinline__ void dyn_step(State<S> &s, S a0, S a1, S a2, S a3, const Consts<U> &c, F fx, F fy, F fz)
{
    const S thrust = (a0 + U(1)) * c.thrust_half;  // quadrotor.py:259, free.py:82
    const S wtx = a1 * c.komega[0], wty = a2 * c.komega[1], wtz = a3 * c.komega[2];  // quadrotor.py:260, free.py:122,82
    dyn_core<S, U, ENTRY_NORM, F>(s, thrust, wtx, wty, wtz, c, fx, fy, fz);
}

template <class S, class U, bool ENTRY_NORM, class F>
__device__ __forceinline__ void dyn_core(State<S> &s, S thrust, S wtx, S wty, S wtz, const Consts<U> &c, F fx, F fy, F fz)
{
    // q = x[3:7] / norm (free.py:88)
    S x = s.qx, y = s.qy, z = s.qz, w = s.qw;
    if (ENTRY_NORM) {
        const S rn = rsqrt_(s.qx * s.qx + s.qy * s.qy + s.qz * s.qz + s.qw * s.qw);
        x = s.qx * rn; y = s.qy * rn; z = s.qz * rn; w = s.qw * rn;
    }
    // Q @ [0,0,T]: third column of qtoQ(q) (geom.py:68-77)
    const S Qz0 = U(2) * (x * z + y * w), Qz1 = U(2) * (y * z - x * w), Qz2 = w * w - x * x - y * y + z * z;
    // v_dot = [0,0,-g] + 1/m (Q [0,0,T] + f)   (free.py:97-99)
    const S vdx = (Qz0 * thrust + fx) * c.inv_m;
    const S vdy = (Qz1 * thrust + fy) * c.inv_m;
    const S vdz = (Qz2 * thrust + fz) * c.inv_m + c.neg_g;
    // q_dot = 0.5 L(q) H omega = 0.5 [w om + v x om, -v.om]   (free.py:96)
    const S qdx = w * s.ox + (y * s.oz - z * s.oy);
    const S qdy = w * s.oy + (z * s.ox - x * s.oz);
    const S qdz = w * s.oz + (x * s.oy - y * s.ox);
    const S qdw = -(x * s.ox + y * s.oy + z * s.oz);
    // explicit Euler with OLD derivatives (free.py:102-107)
    s.px = s.px + s.vx * c.dt;
    s.py = s.py + s.vy * c.dt;
    s.pz = s.pz + s.vz * c.dt;
    s.vx = s.vx + vdx * c.dt;
    s.vy = s.vy + vdy * c.dt;
    s.vz = s.vz + vdz * c.dt;
    const S nx = x + qdx * c.half_dt, ny = y + qdy * c.half_dt, nz = z + qdz * c.half_dt, nw = w + qdw * c.half_dt;
    s.ox = s.ox * c.alpha + wtx * c.one_m_alpha;
    s.oy = s.oy * c.alpha + wty * c.one_m_alpha;
    s.oz = s.oz * c.alpha + wtz * c.one_m_alpha;
    // re-normalise (free.py:139)
    const S rn2 = rsqrt_(nx * nx + ny * ny + nz * nz + nw * nw);
    s.qx = nx * rn2;
    s.qy = ny * rn2;
    s.qz = nz * rn2;
    s.qw = nw * rn2;
}

}  // namespace qm
