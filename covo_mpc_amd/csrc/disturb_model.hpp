// disturb_model.hpp -- the six disturbance models of quadjax/dynamics/free.py:9-72 as device functions (gfx950).
//
// free_dynamics_3d_bodyrate (free.py:114-202) evaluates disturb_func on the PRE-step state and stores the result as the
// f_disturb the NEXT step integrates (free.py:147,91,98):
//   none      0                                                              (free.py:72)
//   gaussian  dyn_noise_scale * normal(key, (3,));  dyn_noise_scale is zeroed by step_env(deterministic=True)  (66-70; quadrotor.py:234)
//   periodic  time % period == 0 ? uniform(key, (3,), -scale, scale) : state.f_disturb                        (10-24)
//   sin       dp[:3] scale sin(2 pi / (dp[:3] period/3 + period) time + 2 pi dp[3:6])                         (27-38)
//   drag      -|scale| rel |rel| / 1.5^2,  rel = vel - dp[:3] / 2                                             (41-47)
//   mixed     (drag + sin + periodic) / 3                                                                     (50-56)
// Random draws follow covo_mpc_amd/random.py (Philox4x32-10 keys, fp64 host formulas rounded to fp32): the same numbers the
// Python env draws for the same key.  Used by env_step.hip, pid_nominal.hip (true env steps) and by disturb_table_kernel
// (disturb.hip), which resolves everything that is wave-uniform per rollout step into a table the rollout / Hessian read.
#pragma once
#include "covo_common.hpp"
#include "rng_device.hpp"

namespace dm {

struct Model {
    int kind, period;
    float scale, noise_scale;
    float dp[6];
};

__host__ __device__ inline Model make_model(const covo_env_params &p)
{
    Model m;
    m.kind = p.disturb_kind;
    m.period = p.disturb_period > 0 ? p.disturb_period : 1;
    m.scale = p.disturb_scale;
    m.noise_scale = p.dyn_noise_scale;
    for (int i = 0; i < 6; ++i) m.dp[i] = p.disturb_params[i];
    return m;
}

// child i of split(key) / element i of uniform(key, (3,), lo, hi) and normal(key, (3,)) exactly as random.py forms them
__device__ __forceinline__ void split(const uint32_t (&key)[2], uint32_t i, uint32_t (&child)[2])
{
    uint32_t r[4];
    rngd::philox4x32_10(i, 0u, 0u, 0x5EEDu, key[0], key[1], r);
    child[0] = r[0];
    child[1] = r[1];
}
// (fp contraction is switched off inside the model functions: they are inlined into several kernels -- the fused step's table
// launch, the env-batched one, covo_disturb_table, the env step -- and left to itself hipcc fuses a different subset of their
// a * b + c per call site, so that the same model gave tables differing in the last ulp)
__device__ __forceinline__ float uniform3(const uint32_t (&key)[2], int i, float lo, float hi)
{
#pragma clang fp contract(off)
    uint32_t b[4];
    rngd::philox4x32_10(0u, 0u, 0u, 0xB175u, key[0], key[1], b);
    const double u = ((double)(b[i] >> 8) + 0.5) / 16777216.0;
    return (float)((double)lo + ((double)hi - (double)lo) * u);
}
__device__ __forceinline__ float normal3(const uint32_t (&key)[2], int i)
{
#pragma clang fp contract(off)
    uint32_t b1[4], b2[4];
    rngd::philox4x32_10(0u, 0u, 0u, 0xB175u, key[0], key[1], b1);
    rngd::philox4x32_10((uint32_t)((3 + i) >> 2), 0u, 0u, 0xB175u, key[0], key[1], b2);
    const double u1 = ((double)(b1[i] >> 8) + 0.5) / 16777216.0;
    const double u2 = ((double)(b2[(3 + i) & 3] >> 8) + 0.5) / 16777216.0;
    return (float)(sqrt(-2.0 * log(u1)) * cos(2.0 * 3.141592653589793 * u2));
}
// disturb_key of step_env(k, ...): raw_step: key, step_key = split(k) (quadrotor.py:262); step_fn: key, key_dyn = split(step_key)
// (free.py:136); disturb_key, key = split(key) (free.py:144)
__device__ __forceinline__ void disturb_key(const uint32_t (&k)[2], uint32_t (&out)[2])
{
    uint32_t a[2], b[2];
    split(k, 1u, a);
    split(a, 0u, b);
    split(b, 0u, out);
}

// component i of free.py:27-38 at `time` (fp32, like the host env)
__device__ __forceinline__ float sin_term(const Model &m, int time, int i)
{
#pragma clang fp contract(off)
    const float scale = m.dp[i] * m.scale;
    const float period = m.dp[i] * (float)((double)m.period / 3.0) + (float)m.period;
    const float phase = m.dp[3 + i] * 6.2831853071795864769f;
    return scale * sinf(6.2831853071795864769f / period * (float)time + phase);
}
// component i of free.py:41-47
__device__ __forceinline__ float drag_term(const Model &m, float vel_i, int i)
{
#pragma clang fp contract(off)
    const float rel = vel_i - m.dp[i] * 0.5f;
    return -fabsf(m.scale) * rel * fabsf(rel) / 2.25f;
}
// multiplier of rel |rel| in the rollout / Hessian kernels: c_drag * (-|scale| / 1.5^2)
__host__ __device__ inline float drag_coeff(const Model &m)
{
    const float c = m.kind == COVO_DISTURB_DRAG ? 1.0f : (m.kind == COVO_DISTURB_MIXED ? 1.0f / 3.0f : 0.0f);
    return -c * (m.scale < 0.0f ? -m.scale : m.scale) / 2.25f;
}

// component i of the next step's f_disturb from the PRE-step state (time, vel_i, f_i) of a TRUE env step; dkey = disturb_key
__device__ __forceinline__ float next_force(const Model &m, const uint32_t (&dkey)[2], int time, float vel_i, float f_i, int i,
                                            bool deterministic)
{
#pragma clang fp contract(off)
    switch (m.kind) {
    case COVO_DISTURB_GAUSSIAN: return deterministic ? 0.0f : m.noise_scale * normal3(dkey, i);
    case COVO_DISTURB_PERIODIC: return (time % m.period == 0) ? uniform3(dkey, i, -m.scale, m.scale) : f_i;
    case COVO_DISTURB_SIN: return sin_term(m, time, i);
    case COVO_DISTURB_DRAG: return drag_term(m, vel_i, i);
    case COVO_DISTURB_MIXED: {
        const float per = (time % m.period == 0) ? uniform3(dkey, i, -m.scale, m.scale) : f_i;
        return (drag_term(m, vel_i, i) + sin_term(m, time, i) + per) / 3.0f;
    }
    default: return 0.0f;
    }
}

}  // namespace dm
