// env_step.hip -- one closed-loop environment step on the device (gfx950): SURVEY.md 8f-1.
//
// Replaces, for the quadrotor env the controllers are evaluated on, one call of
//   Quad3D.step_env (quadjax/envs/quadrotor.py:215-248) = clip -> raw_step (250-263) ->
//   free_dynamics_3d_bodyrate (dynamics/free.py:114-202: Euler step, re-normalised quaternion, the
//   disturbance for the NEXT step from the PRE-step state (free.py:9-72,147: all six models, disturb_model.hpp),
//   time+1, targets = traj[time+1]) -> reward / done of the PRE-step
//   state (243-244) -> get_info (314-361: err_pos/err_vel of the pre-step state, the noisy copy of the NEXT
//   state that the controller plans from).
// The reference runs a whole episode as one XLA program (quadrotor.py:506-591); with this kernel between
// two controller graph replays an episode is 300 x (one async 48-byte copy + one graph + this launch) with
// ONE host sync at the end.  The kernel derives the five noise keys from the step key exactly as the Python
// env does (covo_mpc_amd/envs/quadrotor.py, random.py) and evaluates normal(key, (n,)) with the same
// counter layout: block j = Philox4x32-10(counter (j,0,0,0xB175), key), words b[0..2n), u1 = b[i], u2 = b[n+i],
// z = sqrt(-2 ln u1) cos(2 pi u2) in fp64, rounded to fp32.
// Auto-reset on done (BaseEnvironment.step, quadjax/envs/base.py:22-40; round 6): when the PRE-step state is terminal the kernel
// stores reset_env(key_reset)'s state, noisy copy and NEW reference trajectory instead of the stepped ones (env_reset.hpp), and
// the log row carries the reset state's errors -- what eval_env reads from info after env.step (quadrotor.py:531-538).
#include <cstring>
#include "covo_common.hpp"
#include "disturb_model.hpp"
#include "env_reset.hpp"

struct EnvStepArgs {
    float *state;          // [32] true state, updated in place
    float *noisy;          // [32] the noisy copy of the new state (what the controller reads)
    float *pos_traj, *vel_traj, *acc_traj;  // [T][3]; rewritten by an auto-reset
    const float *action;   // [4] (device): the controller's u = a_mean[0]
    float *log;            // [max_steps][4]: reward, err_pos, err_vel, done of the PRE-step state (nullable)
    int T, log_index, noisy_on;
    float obs_noise_scale;
    dm::Model dist;        // the disturbance model (free.py:9-72)
    int reward;            // COVO_REWARD_*
    uint32_t step_key[2];  // the key Quad3D.step receives; the five noise keys are derived from it on the device
    qm::Consts<float> c;
    int max_steps;
    int rollover;  // is_terminal's rollover test (quadrotor.py:486-490)
    er::ResetArgs reset;   // kind == COVO_TRAJ_NONE: no auto-reset
};

// element i of normal(key, (n,)) -- host formula (random.py) in fp64
__device__ __forceinline__ float host_normal(const uint32_t (&key)[2], int n, int i)
{
    uint32_t b1[4], b2[4];
    rngd::philox4x32_10((uint32_t)(i >> 2), 0u, 0u, 0xB175u, key[0], key[1], b1);
    rngd::philox4x32_10((uint32_t)((n + i) >> 2), 0u, 0u, 0xB175u, key[0], key[1], b2);
    const double u1 = ((double)(b1[i & 3] >> 8) + 0.5) / 16777216.0;
    const double u2 = ((double)(b2[(n + i) & 3] >> 8) + 0.5) / 16777216.0;
    return (float)(sqrt(-2.0 * log(u1)) * cos(2.0 * 3.141592653589793 * u2));
}

// child i of split(key, num) (random.py: Philox counter (i, 0, 0, 0x5EED), first two words)
__device__ __forceinline__ void host_split(const uint32_t (&key)[2], uint32_t i, uint32_t (&child)[2])
{
    uint32_t r[4];
    rngd::philox4x32_10(i, 0u, 0u, 0x5EEDu, key[0], key[1], r);
    child[0] = r[0];
    child[1] = r[1];
}

__device__ __forceinline__ void env_step_body(const EnvStepArgs &A)
{
    __shared__ float z[16];
    __shared__ float sst[COVO_STATE_FLOATS];   // the state as loaded, then as stepped
    __shared__ float sact[4];
    __shared__ float straj[2][9];              // pos / vel / acc targets of row `guess` (speculative) and of row time + 1
    __shared__ double skp[16][3];              // an auto-reset's zigzag key points
    const int lane = threadIdx.x;
    // One launch of one wave is latency: everything that does not depend on anything is requested first -- the state (lanes
    // 0..31, one coalesced load), the action, and the trajectory row the NEXT state will point at, guessed from the step count
    // (time = steps since reset for every episode the drivers run; checked against the state's own clock below and re-read if
    // wrong) -- and lands while lanes 0..15 hash the step's noise.  (Round 1: lane 0 alone, state -> time -> row -> state read
    // back for the noisy copy: four dependent round trips, 9.5 us.)
    float *__restrict__ st = A.state;
    const int guess = A.log_index + 1 < 0 ? 0 : (A.log_index + 1 > A.T - 1 ? A.T - 1 : A.log_index + 1);
    float ld_state = 0.0f, ld_act = 0.0f, ld_traj = 0.0f;
    if (lane < COVO_STATE_FLOATS) ld_state = st[lane];
    if (lane >= 32 && lane < 36) ld_act = A.action[lane - 32];
    if (lane >= 40 && lane < 49) {
        const int q = lane - 40;
        const float *__restrict__ tr = q < 3 ? A.pos_traj : (q < 6 ? A.vel_traj : A.acc_traj);
        ld_traj = tr[3 * guess + q % 3];
    }
    // ---- the 16 normals of the step: disturb(3) pos(3) vel(3) quat(4) omega(3).  Key tree of Quad3D.step(key, ...):
    //   k = split(key)[0]                       base.py:22 (child 1 is the reset key)
    //   info_key = split(k)[0], raw = split(k)[1]      quadrotor.py:246 / :262 (both split the SAME key)
    //   disturb = split(split(raw)[0])[0]       free.py:136 (key, key_dyn), :144 (disturb_key, key)
    //   pos, vel, quat, omega = split(info_key, 5)[0..3]   quadrotor.py:324
    if (lane < 16) {
        const int grp = lane < 3 ? 0 : lane < 6 ? 1 : lane < 9 ? 2 : lane < 13 ? 3 : 4;
        const int base = grp == 0 ? 0 : grp == 1 ? 3 : grp == 2 ? 6 : grp == 3 ? 9 : 13;
        const int n = grp == 3 ? 4 : 3;
        const uint32_t sk[2] = {A.step_key[0], A.step_key[1]};
        uint32_t k[2], t[2], key[2];
        host_split(sk, 0u, k);
        if (grp == 0) {
            host_split(k, 1u, t);     // raw_step's step_key
            host_split(t, 0u, key);   // step_fn: key
            host_split(key, 0u, t);   // disturb_key
            key[0] = t[0];
            key[1] = t[1];
        } else {
            host_split(k, 0u, t);                      // info_key
            host_split(t, (uint32_t)(grp - 1), key);   // rng_pos / vel / quat / omega
        }
        // lanes 0..2: the disturbance model's own draw for this step -- the gaussian model's normal, or the uniform periodic /
        // mixed redraw with when time % period == 0 (free.py:16-21; same disturb_key); lanes 3..15 the observation noise
        if (grp == 0 && (A.dist.kind == COVO_DISTURB_PERIODIC || A.dist.kind == COVO_DISTURB_MIXED))
            z[lane] = dm::uniform3(key, lane, -A.dist.scale, A.dist.scale);
        else
            z[lane] = host_normal(key, n, lane - base);
    }
    if (lane < COVO_STATE_FLOATS) sst[lane] = ld_state;
    if (lane >= 32 && lane < 36) sact[lane - 32] = ld_act;
    if (lane >= 40 && lane < 49) straj[0][lane - 40] = ld_traj;
    __syncthreads();
    const int time = __float_as_int(sst[ST_TIME]);
    const int t1 = time + 1;
    const int idx = t1 < 0 ? 0 : (t1 > A.T - 1 ? A.T - 1 : t1);  // JAX gather clamps (free.py:150-155)
    const bool reread = idx != guess;  // uniform
    if (reread && lane >= 40 && lane < 49) {
        const int q = lane - 40;
        const float *__restrict__ tr = q < 3 ? A.pos_traj : (q < 6 ? A.vel_traj : A.acc_traj);
        straj[1][q] = tr[3 * idx + q % 3];
    }
    // ---- termination of the PRE-step state (quadrotor.py:244, 479-490): every lane evaluates it on the same LDS words
    bool done = (time >= A.max_steps) ||
                fmaxf(fmaxf(fabsf(sst[ST_POS + 0]), fabsf(sst[ST_POS + 1])), fabsf(sst[ST_POS + 2])) > A.c.pos_limit;
    if (A.rollover)  // quadrotor.py:486-490
        done = done || sst[ST_QUAT + 3] < 0.70710678118654752f ||
               fmaxf(fmaxf(fabsf(sst[ST_OMEGA + 0]), fabsf(sst[ST_OMEGA + 1])), fabsf(sst[ST_OMEGA + 2])) > 100.0f;
    const bool reset = done && A.reset.kind != COVO_TRAJ_NONE;  // uniform
    float r_pre = 0.0f;
    if (lane == 0) {
        qm::State<float> s;
        s.px = sst[ST_POS + 0]; s.py = sst[ST_POS + 1]; s.pz = sst[ST_POS + 2];
        s.vx = sst[ST_VEL + 0]; s.vy = sst[ST_VEL + 1]; s.vz = sst[ST_VEL + 2];
        s.qx = sst[ST_QUAT + 0]; s.qy = sst[ST_QUAT + 1]; s.qz = sst[ST_QUAT + 2]; s.qw = sst[ST_QUAT + 3];
        s.ox = sst[ST_OMEGA + 0]; s.oy = sst[ST_OMEGA + 1]; s.oz = sst[ST_OMEGA + 2];
        const float fx = sst[ST_FDIST + 0], fy = sst[ST_FDIST + 1], fz = sst[ST_FDIST + 2];
        const float tx = sst[ST_POSTAR + 0], ty = sst[ST_POSTAR + 1], tz = sst[ST_POSTAR + 2];
        const float tvx = sst[ST_VELTAR + 0], tvy = sst[ST_VELTAR + 1], tvz = sst[ST_VELTAR + 2];
        // ---- reward / errors of the PRE-step state (quadrotor.py:243, 352-353; utils.py:285-294)
        if (A.log != nullptr) {
            r_pre = qm::reward_kind<float, float>(A.reward, s, tx, ty, tz, tvx, tvy, tvz);
            if (!reset) {
                const float ex = tx - s.px, ey = ty - s.py, ez = tz - s.pz;
                const float wx = tvx - s.vx, wy = tvy - s.vy, wz = tvz - s.vz;
                float *__restrict__ lg = A.log + 4 * A.log_index;
                lg[0] = r_pre;
                lg[1] = sqrtf(ex * ex + ey * ey + ez * ez);
                lg[2] = sqrtf(wx * wx + wy * wy + wz * wz);
                lg[3] = done ? 1.0f : 0.0f;
            }
        }
        if (!reset) {
        // ---- one Euler step with the state's current disturbance (free.py:91,98)
        const float a0 = qm::clip11_(qm::clip11_(sact[0])), a1 = qm::clip11_(qm::clip11_(sact[1]));
        const float a2 = qm::clip11_(qm::clip11_(sact[2])), a3 = qm::clip11_(qm::clip11_(sact[3]));
        // the next step's force from the PRE-step state (free.py:147)
        float fn[3];
        {
            const float vel[3] = {s.vx, s.vy, s.vz}, fcur[3] = {fx, fy, fz};
            const dm::Model &m = A.dist;
            const bool hit = (time % m.period) == 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float per = hit ? z[i] : fcur[i];
                switch (m.kind) {
                case COVO_DISTURB_GAUSSIAN: fn[i] = m.noise_scale * z[i]; break;  // free.py:66-70
                case COVO_DISTURB_PERIODIC: fn[i] = per; break;
                case COVO_DISTURB_SIN: fn[i] = dm::sin_term(m, time, i); break;
                case COVO_DISTURB_DRAG: fn[i] = dm::drag_term(m, vel[i], i); break;
                case COVO_DISTURB_MIXED: fn[i] = (dm::drag_term(m, vel[i], i) + dm::sin_term(m, time, i) + per) / 3.0f; break;
                default: fn[i] = 0.0f; break;
                }
            }
        }
        qm::dyn_step<float, float>(s, a0, a1, a2, a3, A.c, fx, fy, fz);
        sst[ST_POS + 0] = s.px; sst[ST_POS + 1] = s.py; sst[ST_POS + 2] = s.pz;
        sst[ST_VEL + 0] = s.vx; sst[ST_VEL + 1] = s.vy; sst[ST_VEL + 2] = s.vz;
        sst[ST_QUAT + 0] = s.qx; sst[ST_QUAT + 1] = s.qy; sst[ST_QUAT + 2] = s.qz; sst[ST_QUAT + 3] = s.qw;
        sst[ST_OMEGA + 0] = s.ox; sst[ST_OMEGA + 1] = s.oy; sst[ST_OMEGA + 2] = s.oz;
        sst[ST_FDIST + 0] = fn[0]; sst[ST_FDIST + 1] = fn[1]; sst[ST_FDIST + 2] = fn[2];
        sst[ST_TIME] = __int_as_float(t1);
        }
    }
    if (reset) {
        // ---- base.py:26-40: select(done, reset_env(key_reset), stepped) on state and info; the controller's state carries on
        __syncthreads();  // lane 0 has read the pre-step state
        const uint32_t sk[2] = {A.step_key[0], A.step_key[1]};
        er::reset_env(A.reset, sk, A.pos_traj, A.vel_traj, A.acc_traj, sst, z, straj[1], skp);
        __syncthreads();
        if (lane == 0 && A.log != nullptr) {  // info_re of get_info(info_key, state_re, state_re): the RESET state's errors
            const float ex = sst[ST_POSTAR + 0] - sst[ST_POS + 0], ey = sst[ST_POSTAR + 1] - sst[ST_POS + 1],
                        ez = sst[ST_POSTAR + 2] - sst[ST_POS + 2];
            const float wx = sst[ST_VELTAR + 0] - sst[ST_VEL + 0], wy = sst[ST_VELTAR + 1] - sst[ST_VEL + 1],
                        wz = sst[ST_VELTAR + 2] - sst[ST_VEL + 2];
            float *__restrict__ lg = A.log + 4 * A.log_index;
            lg[0] = r_pre;
            lg[1] = sqrtf(ex * ex + ey * ey + ez * ez);
            lg[2] = sqrtf(wx * wx + wy * wy + wz * wz);
            lg[3] = 1.0f;
        }
    }
    __syncthreads();  // also: the re-read row (if any) has landed in straj[1] -- its writers waited for their own loads
    if (lane < 9 && !reset) {
        const float v = straj[reread ? 1 : 0][lane];
        sst[(lane < 3 ? ST_POSTAR : (lane < 6 ? ST_VELTAR : ST_ACCTAR)) + lane % 3] = v;
    }
    __syncthreads();
    // ---- the new state, and the noisy copy the controller plans from (quadrotor.py:322-350; quaternion NOT re-normalised)
    if (lane < COVO_STATE_FLOATS) {
        const float v = sst[lane];
        st[lane] = v;
        // noise = (z * scale) * c, then state + noise: three separately rounded operations as in quadrotor.py:322-350 (no FMA
        // contraction: the copy then equals the Python env's bit for bit given the same true state)
        const float on = A.noisy_on ? A.obs_noise_scale : 0.0f;
        float zz = 0.0f, cc = 0.0f;
        bool noisy_field = true;
        if (lane >= ST_POS && lane < ST_POS + 3) { zz = z[3 + lane - ST_POS]; cc = 0.25f; }
        else if (lane >= ST_VEL && lane < ST_VEL + 3) { zz = z[6 + lane - ST_VEL]; cc = 0.5f; }
        else if (lane >= ST_QUAT && lane < ST_QUAT + 4) { zz = z[9 + lane - ST_QUAT]; cc = 0.02f; }
        else if (lane >= ST_OMEGA && lane < ST_OMEGA + 3) { zz = z[13 + lane - ST_OMEGA]; cc = 0.5f; }
        else noisy_field = false;
        float nz = (zz * on) * cc;
        asm volatile("" : "+v"(nz));  // keeps hipcc from contracting the product into the add below (__fadd_rn / __fmul_rn do not)
        A.noisy[lane] = noisy_field ? v + nz : v;
    }
}

__global__ __launch_bounds__(64) void env_step_kernel(const EnvStepArgs A) { env_step_body(A); }

// ---- E env instances in one launch (BASELINE configs[4]; quadrotor.py:132-171 samples each instance's parameters, :506-591
// drives it): workgroup e steps instance e -- its own true state, noisy copy, trajectories, action (the first four entries of
// its mean), model constants and key.  Same body as the single-instance kernel, so instance e's step is bit-identical to
// covo_env_step on that instance alone.
struct EnvInst {            // what domain randomisation varies per instance (device array, built once per episode)
    qm::Consts<float> c;
    dm::Model dist;
    double reset_dt, reset_disturb_scale;
};
struct EnvStepBatchKeys {
    uint32_t k[COVO_MAX_ENVS][2];  // the key Quad3D.step receives, per instance
};
struct EnvStepBatchArgs {
    float *states;         // [E][32] true states, updated in place
    float *noisy;          // [E][32] the noisy copies (the batched controller's `states`)
    float *pos_traj, *vel_traj, *acc_traj;  // [E][T][3]; an instance's block is rewritten by its auto-reset
    const float *a_mean;   // [E][128]: instance e's action = a_mean[e][0..3]
    float *log;            // [E][log_stride][4] or null
    const EnvInst *inst;   // [E]
    int T, log_index, log_stride, noisy_on;
    float obs_noise_scale;
    int reward, max_steps, rollover, reset_traj;
};
__global__ __launch_bounds__(64) void env_step_batched_kernel(const EnvStepBatchArgs B, const EnvStepBatchKeys K)
{
    const int e = blockIdx.x;
    EnvStepArgs A;
    A.state = B.states + (size_t)e * COVO_STATE_FLOATS;
    A.noisy = B.noisy + (size_t)e * COVO_STATE_FLOATS;
    A.pos_traj = B.pos_traj + (size_t)e * B.T * 3;
    A.vel_traj = B.vel_traj + (size_t)e * B.T * 3;
    A.acc_traj = B.acc_traj + (size_t)e * B.T * 3;
    A.action = B.a_mean + (size_t)e * COVO_NA;
    A.log = B.log ? B.log + (size_t)e * B.log_stride * 4 : nullptr;
    A.T = B.T;
    A.log_index = B.log_index;
    A.noisy_on = B.noisy_on;
    A.obs_noise_scale = B.obs_noise_scale;
    A.dist = B.inst[e].dist;
    A.reward = B.reward;
    A.step_key[0] = K.k[e][0];
    A.step_key[1] = K.k[e][1];
    A.c = B.inst[e].c;
    A.max_steps = B.max_steps;
    A.rollover = B.rollover;
    A.reset.kind = B.reset_traj;
    A.reset.max_steps = B.max_steps;
    A.reset.T = B.T;
    A.reset.dt = B.inst[e].reset_dt;
    A.reset.disturb_scale = B.inst[e].reset_disturb_scale;
    env_step_body(A);
}

size_t env_step_inst_bytes(int n) { return (size_t)n * sizeof(EnvInst); }
void env_step_fill_inst(const covo_env_params *params, int n, void *out)
{
    EnvInst *o = reinterpret_cast<EnvInst *>(out);
    for (int i = 0; i < n; ++i) {
        o[i].c = make_consts<float>(params[i]);
        o[i].dist = dm::make_model(params[i]);
        o[i].reset_dt = params[i].reset_dt;
        o[i].reset_disturb_scale = params[i].reset_disturb_scale;
    }
}

// the auto-reset's preconditions (covo_hip.h: T is the generator's row count)
static int check_reset(const covo_env_params &p, int T, const char *what)
{
    if (p.reset_traj == COVO_TRAJ_NONE) return 0;
    const int rows = er::traj_rows(p.reset_traj, p.max_steps_in_episode);
    if (rows < 0 || rows != T || p.max_steps_in_episode / 40 + 2 > 16 || !(p.reset_dt > 0.0) || !(p.reset_disturb_scale >= 0.0)) {
        covo_set_error("%s: auto-reset (reset_traj=%d) needs T = %d rows for max_steps_in_episode=%d (got T=%d), "
                       "max_steps_in_episode < 600, reset_dt > 0 (%g) and reset_disturb_scale >= 0 (%g)",
                       what, p.reset_traj, rows, p.max_steps_in_episode, T, p.reset_dt, p.reset_disturb_scale);
        return COVO_E_BADARG;
    }
    return 0;
}

// params0: what all instances share (reward, max_steps, rollover switch, reset generator); the per-instance constants come from `inst`
int launch_env_step_batched(float *states, float *noisy, const float *pos_traj, const float *vel_traj, const float *acc_traj, int T,
                            const covo_env_params &params0, const void *inst_dev, int E, const float *a_mean,
                            const uint32_t *step_keys /* host [E][2] */, int noisy_on, float obs_noise_scale, float *log,
                            int log_stride, int log_index, hipStream_t s)
{
    if (E <= 0 || E > COVO_MAX_ENVS) {
        covo_set_error("env_step_batched: n_envs=%d outside (0, %d]", E, COVO_MAX_ENVS);
        return COVO_E_BADARG;
    }
    if (int rc = check_reset(params0, T, "env_step_batched")) return rc;
    EnvStepBatchArgs B;
    B.states = states;
    B.noisy = noisy;
    B.pos_traj = const_cast<float *>(pos_traj);
    B.vel_traj = const_cast<float *>(vel_traj);
    B.acc_traj = const_cast<float *>(acc_traj);
    B.reset_traj = params0.reset_traj;
    B.a_mean = a_mean;
    B.log = log;
    B.inst = reinterpret_cast<const EnvInst *>(inst_dev);
    B.T = T;
    B.log_index = log_index;
    B.log_stride = log_stride;
    B.noisy_on = noisy_on;
    B.obs_noise_scale = obs_noise_scale;
    B.reward = params0.reward_kind;
    B.max_steps = params0.max_steps_in_episode;
    B.rollover = params0.rollover_terminate != 0;
    EnvStepBatchKeys K;
    std::memset(&K, 0, sizeof(K));
    for (int e = 0; e < E; ++e) {
        K.k[e][0] = step_keys[2 * e];
        K.k[e][1] = step_keys[2 * e + 1];
    }
    hipLaunchKernelGGL(env_step_batched_kernel, dim3(E), dim3(64), 0, s, B, K);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_env_step(float *state, float *noisy, const float *pos_traj, const float *vel_traj, const float *acc_traj, int T,
                    const covo_env_params &p, const float *action, const uint32_t *step_key, int noisy_on,
                    float obs_noise_scale, float *log, int log_index, hipStream_t s)
{
    if (int rc = check_reset(p, T, "env_step")) return rc;
    EnvStepArgs A;
    A.state = state;
    A.noisy = noisy;
    A.pos_traj = const_cast<float *>(pos_traj);
    A.vel_traj = const_cast<float *>(vel_traj);
    A.acc_traj = const_cast<float *>(acc_traj);
    A.reset.kind = p.reset_traj;
    A.reset.max_steps = p.max_steps_in_episode;
    A.reset.T = T;
    A.reset.dt = p.reset_dt;
    A.reset.disturb_scale = p.reset_disturb_scale;
    A.action = action;
    A.log = log;
    A.T = T;
    A.log_index = log_index;
    A.noisy_on = noisy_on;
    A.obs_noise_scale = obs_noise_scale;
    A.dist = dm::make_model(p);
    A.reward = p.reward_kind;
    A.step_key[0] = step_key[0];
    A.step_key[1] = step_key[1];
    A.c = make_consts<float>(p);
    A.max_steps = p.max_steps_in_episode;
    A.rollover = p.rollover_terminate != 0;
    hipLaunchKernelGGL(env_step_kernel, dim3(1), dim3(64), 0, s, A);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}
