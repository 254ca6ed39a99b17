// disturb.hip -- the per-step disturbance table of a rollout (gfx950): covo_disturb_table.
//
// Inside a controller every rollout step k >= 1 integrates the force disturb_func returned for the PRE-step state of step
// k-1 (quadjax/dynamics/free.py:147,91,98).  Four of the six models (free.py:9-72) never look at the sample: none, gaussian,
// periodic (one uniform draw or the held vector), sin (time only) -- their whole horizon is 32 wave-uniform vectors, resolved
// here once per control step so that the N x H rollout and the Hessian spend nothing per sample on them.  drag depends on the
// sample's own velocity and mixed = (drag + sin + periodic) / 3 also on its own previous force; for those the table carries
// the wave-uniform remainder:
//     row k = {g_k[3], c_k}:   f_k = c_drag drag(vel_{k-1}) + c_k f_{k-1} + g_k          (k = 1 .. H-1; row 0 is unused:
//     step 0 integrates the state's own f_disturb)
// with c_drag = 1 (drag), 1/3 (mixed), 0 otherwise (dm::drag_coeff) and c_k = 1/3 while mixed's periodic part holds.
// Keys: who calls step_env with which key decides which uniform a redraw step sees (COVO_DISTURB_KEYS_*, include/covo_hip.h).
#include "disturb_model.hpp"

struct DisturbTableArgs {
    const float *state;      // [batch][COVO_STATE_FLOATS]
    const uint32_t *keys;    // [batch][2] or null
    uint32_t key[2];         // when keys == null
    float4 *out;             // [batch][COVO_H]
    int batch, key_mode, deterministic;
    dm::Model m;
};

// One WAVE per table, lane k = row k + 1 (round 3; the first version walked the 31 rows on one lane: 25-43 us per control step
// of Philox splits and sinf -- more than the Hessian).  Only the key chain of the per-step modes is serial (split after split),
// and it is walked by all lanes together only as far as the last row that draws: a horizon holds at most ceil(H / period)
// redraw steps.  periodic's hold (g_k = f_{k-1} when the step does not redraw) resolves from the ballot of the redraw lanes.
__device__ __forceinline__ void disturb_table_rows_wave(const dm::Model &m, const float *__restrict__ st, const uint32_t (&key0)[2],
                                                        int key_mode, bool deterministic, float4 *__restrict__ out)
{
    const int k = threadIdx.x & 63;
    const bool act = k < COVO_H - 1;
    const int time = __float_as_int(st[ST_TIME]) + k;
    const bool hit = act && (time % m.period) == 0;
    const bool need_draw = act && ((m.kind == COVO_DISTURB_GAUSSIAN && !deterministic) ||
                                   ((m.kind == COVO_DISTURB_PERIODIC || m.kind == COVO_DISTURB_MIXED) && hit));
    // the key step_env receives at rollout step k
    uint32_t sk[2] = {key0[0], key0[1]};
    const unsigned long long need = __ballot(need_draw);
    if (key_mode != COVO_DISTURB_KEYS_SHARED && need != 0ull) {
        const int last = 63 - __builtin_clzll(need);
        uint32_t key[2] = {key0[0], key0[1]}, mine[2] = {key0[0], key0[1]};
        for (int j = 0; j <= last; ++j) {  // (uniform)
            uint32_t nk[2];
            if (key_mode == COVO_DISTURB_KEYS_HESSIAN) {  // rng_act, key = split(key) (covo.py:151)
                if (j == k) { mine[0] = key[0]; mine[1] = key[1]; }
                dm::split(key, 1u, nk);
            } else {  // rng_act, key = split(key); rng_step, key = split(key) (covo.py:60,66)
                uint32_t k1[2];
                dm::split(key, 1u, k1);
                if (j == k) { mine[0] = k1[0]; mine[1] = k1[1]; }
                dm::split(k1, 1u, nk);
            }
            key[0] = nk[0]; key[1] = nk[1];
        }
        dm::split(mine, 0u, sk);
    }
    uint32_t dk[2] = {0u, 0u};
    if (need_draw) dm::disturb_key(sk, dk);
    float draw[3] = {0.0f, 0.0f, 0.0f};
    if (need_draw) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            draw[i] = (m.kind == COVO_DISTURB_GAUSSIAN) ? m.noise_scale * dm::normal3(dk, i) : dm::uniform3(dk, i, -m.scale, m.scale);
    }
    float g[3] = {0.0f, 0.0f, 0.0f}, c = 0.0f;
    if (m.kind == COVO_DISTURB_PERIODIC) {
        // the latest redraw at or before this row, else the state's own force (free.py:10-24)
        const unsigned long long hm = __ballot(hit);
        const unsigned long long upto = hm & ((k >= 63) ? ~0ull : ((2ull << k) - 1ull));
        const int src = upto ? 63 - __builtin_clzll(upto) : 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float held = __shfl(draw[i], src);
            g[i] = upto ? held : st[ST_FDIST + i];
        }
    } else if (m.kind == COVO_DISTURB_GAUSSIAN) {
#pragma unroll
        for (int i = 0; i < 3; ++i) g[i] = draw[i];
    } else if (m.kind == COVO_DISTURB_SIN) {
#pragma unroll
        for (int i = 0; i < 3; ++i) g[i] = act ? dm::sin_term(m, time, i) : 0.0f;
    } else if (m.kind == COVO_DISTURB_MIXED) {
#pragma unroll
        for (int i = 0; i < 3; ++i) g[i] = act ? (dm::sin_term(m, time, i) + draw[i]) / 3.0f : 0.0f;  // draw = 0 unless the step redraws
        c = hit ? 0.0f : 1.0f / 3.0f;
    }  // none, drag: zeros
    if (act) out[k + 1] = make_float4(g[0], g[1], g[2], c);
    if (k == 0) out[0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

__global__ __launch_bounds__(64) void disturb_table_kernel(const DisturbTableArgs A)
{
    const int b = blockIdx.x;
    const uint32_t key[2] = {A.keys ? A.keys[2 * b] : A.key[0], A.keys ? A.keys[2 * b + 1] : A.key[1]};
    disturb_table_rows_wave(A.m, A.state + (size_t)b * COVO_STATE_FLOATS, key, A.key_mode, A.deterministic != 0,
                            A.out + (size_t)b * COVO_H);
}

// fused step (step.hip): both tables of one control step in one launch -- workgroup 0 the rollouts' (shared step key
// = split(split(rng_act)[0])[1], covo.py:212,225 / mppi.py:53,69), workgroup 1 the Hessian's (per-step keys from the
// raw rng_act, covo.py:39,150-153); dyn = the step's device block {.., raw rng_act at [10], [11]} (step.hip: DynBlock)
__global__ __launch_bounds__(64) void disturb_tables_step_kernel(const float *__restrict__ state, const uint32_t *__restrict__ dyn,
                                                                 dm::Model m, int rollout_deterministic, float4 *__restrict__ tab_rollout,
                                                                 float4 *__restrict__ tab_hess)
{
    const uint32_t raw[2] = {dyn[10], dyn[11]};
    if (blockIdx.x == 0) {
        uint32_t rng1[2], step_key[2];
        dm::split(raw, 0u, rng1);
        dm::split(rng1, 1u, step_key);
        disturb_table_rows_wave(m, state, step_key, COVO_DISTURB_KEYS_SHARED, rollout_deterministic != 0, tab_rollout);
    } else if (tab_hess != nullptr) {
        disturb_table_rows_wave(m, state, raw, COVO_DISTURB_KEYS_HESSIAN, true, tab_hess);
    }
}

// env-batched step (step.hip: covo_mpc_step_batched): the same two tables for every instance -- workgroup (e, 0) the rollouts',
// (e, 1) the Hessian's; instance e has its own state, raw key (dyn[12 e + 10..11]) and, under domain randomisation, its own
// disturb_params (quadrotor.py:152)
__global__ __launch_bounds__(64) void disturb_tables_batched_kernel(const float *__restrict__ states, const uint32_t *__restrict__ dyn,
                                                                    const dm::Model *__restrict__ models, int rollout_deterministic,
                                                                    float4 *__restrict__ tab_rollout, float4 *__restrict__ tab_hess)
{
    const int e = blockIdx.x;
    const dm::Model m = models[e];
    const float *st = states + (size_t)e * COVO_STATE_FLOATS;
    const uint32_t raw[2] = {dyn[12 * e + 10], dyn[12 * e + 11]};
    if (blockIdx.y == 0) {
        uint32_t rng1[2], step_key[2];
        dm::split(raw, 0u, rng1);
        dm::split(rng1, 1u, step_key);
        disturb_table_rows_wave(m, st, step_key, COVO_DISTURB_KEYS_SHARED, rollout_deterministic != 0, tab_rollout + (size_t)e * COVO_H);
    } else {
        disturb_table_rows_wave(m, st, raw, COVO_DISTURB_KEYS_HESSIAN, true, tab_hess + (size_t)e * COVO_H);
    }
}

size_t disturb_models_bytes(int n) { return (size_t)n * sizeof(dm::Model); }
void disturb_fill_models(const covo_env_params *params, int n, void *out)
{
    dm::Model *m = reinterpret_cast<dm::Model *>(out);
    for (int i = 0; i < n; ++i) m[i] = dm::make_model(params[i]);
}

int launch_disturb_tables_batched(const void *models_dev, const float *states, const uint32_t *dyn, int n_envs,
                                  int rollout_deterministic, float *tab_rollout, float *tab_hess, hipStream_t s)
{
    hipLaunchKernelGGL(disturb_tables_batched_kernel, dim3(n_envs, 2), dim3(64), 0, s, states, dyn,
                       reinterpret_cast<const dm::Model *>(models_dev), rollout_deterministic,
                       reinterpret_cast<float4 *>(tab_rollout), reinterpret_cast<float4 *>(tab_hess));
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_disturb_table(const covo_env_params &p, const float *state, int batch, const uint32_t *keys_dev, uint32_t key0,
                         uint32_t key1, int key_mode, int deterministic, float *out, hipStream_t s)
{
    DisturbTableArgs A;
    A.state = state;
    A.keys = keys_dev;
    A.key[0] = key0;
    A.key[1] = key1;
    A.out = reinterpret_cast<float4 *>(out);
    A.batch = batch;
    A.key_mode = key_mode;
    A.deterministic = deterministic;
    A.m = dm::make_model(p);
    hipLaunchKernelGGL(disturb_table_kernel, dim3(batch), dim3(64), 0, s, A);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_disturb_tables_step(const covo_env_params &p, const float *state, const uint32_t *dyn, int rollout_deterministic,
                               float *tab_rollout, float *tab_hess, hipStream_t s)
{
    hipLaunchKernelGGL(disturb_tables_step_kernel, dim3(tab_hess ? 2 : 1), dim3(64), 0, s, state, dyn, dm::make_model(p),
                       rollout_deterministic, reinterpret_cast<float4 *>(tab_rollout), reinterpret_cast<float4 *>(tab_hess));
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}
