// pid_nominal.hip -- the nominal trajectory covo-offline expands its Sigma table around (gfx950).
//
// Replaces the host loop of quadjax/controllers/covo.py:58-99 (get_single_a_cov_offline under lax.scan): from the
// reset state, max_steps_in_episode PID-tracked env steps (NON-deterministic: each draws the gaussian disturbance
// of the next step), and from every one of those states H deterministic PID steps whose actions are the nominal
// mean the Hessian is taken at.  PID law: quadjax/controllers/pid.py:38-84 with the expansion gains of
// covo.py:48-53 (Ki = 0, so the integral state is inert); env step: quadrotor.py:215-263 + free.py:114-202 on the
// shared model (quad_model.hpp).  In Python/numpy this is 300 x 33 scalar env steps = 1.5 s per episode reset;
// here: one lane walks the 300-step chain (keys split exactly as covo_mpc_amd/random.py does, 2 ms), then one
// lane per start state rolls its H nominal steps.
#include "covo_common.hpp"
#include "disturb_model.hpp"

struct PidArgs {
    const float *state0;  // [32]
    const float *pos_traj, *vel_traj, *acc_traj;  // [T][3]
    float *states;        // [n_steps][32]
    float *a_means;       // [n_steps][128]
    int T, n_steps;
    uint32_t key[2];
    qm::Consts<float> c;  // the env's parameters (possibly domain-randomised)
    float pid_m, pid_g, pid_max_thrust, pid_max_omega[3];  // pid.py:33: the PID law uses the DEFAULT parameters
    float Kp, Kd, Kp_att;
    dm::Model dist;       // the disturbance model (free.py:9-72); its gaussian part is off in the deterministic nominal steps
    uint32_t *keys;       // [n_steps][2] the scan's carry key at every start state (nullable for the chain; the nominal kernel
                          // reads it back when the model draws)
};

struct PidState {
    qm::State<float> s;
    float f[3], ptar[3], vtar[3], atar[3];
    int time;
};

__device__ __forceinline__ void pid_split(const uint32_t (&key)[2], uint32_t i, uint32_t (&child)[2])
{
    uint32_t r[4];
    rngd::philox4x32_10(i, 0u, 0u, 0x5EEDu, key[0], key[1], r);
    child[0] = r[0];
    child[1] = r[1];
}
__device__ __forceinline__ float pid_normal3(const uint32_t (&key)[2], int i)
{
    uint32_t b1[4], b2[4];
    rngd::philox4x32_10(0u, 0u, 0u, 0xB175u, key[0], key[1], b1);
    rngd::philox4x32_10((uint32_t)((3 + i) >> 2), 0u, 0u, 0xB175u, key[0], key[1], b2);
    const double u1 = ((double)(b1[i] >> 8) + 0.5) / 16777216.0;
    const double u2 = ((double)(b2[(3 + i) & 3] >> 8) + 0.5) / 16777216.0;
    return (float)(sqrt(-2.0 * log(u1)) * cos(2.0 * 3.141592653589793 * u2));
}

__device__ __forceinline__ void pid_load(PidState &p, const float *__restrict__ st)
{
    p.s.px = st[ST_POS + 0]; p.s.py = st[ST_POS + 1]; p.s.pz = st[ST_POS + 2];
    p.s.vx = st[ST_VEL + 0]; p.s.vy = st[ST_VEL + 1]; p.s.vz = st[ST_VEL + 2];
    p.s.qx = st[ST_QUAT + 0]; p.s.qy = st[ST_QUAT + 1]; p.s.qz = st[ST_QUAT + 2]; p.s.qw = st[ST_QUAT + 3];
    p.s.ox = st[ST_OMEGA + 0]; p.s.oy = st[ST_OMEGA + 1]; p.s.oz = st[ST_OMEGA + 2];
    for (int i = 0; i < 3; ++i) {
        p.f[i] = st[ST_FDIST + i];
        p.ptar[i] = st[ST_POSTAR + i];
        p.vtar[i] = st[ST_VELTAR + i];
        p.atar[i] = st[ST_ACCTAR + i];
    }
    p.time = __float_as_int(st[ST_TIME]);
}
__device__ __forceinline__ void pid_store(const PidState &p, float *__restrict__ st)
{
    st[ST_POS + 0] = p.s.px; st[ST_POS + 1] = p.s.py; st[ST_POS + 2] = p.s.pz;
    st[ST_VEL + 0] = p.s.vx; st[ST_VEL + 1] = p.s.vy; st[ST_VEL + 2] = p.s.vz;
    st[ST_QUAT + 0] = p.s.qx; st[ST_QUAT + 1] = p.s.qy; st[ST_QUAT + 2] = p.s.qz; st[ST_QUAT + 3] = p.s.qw;
    st[ST_OMEGA + 0] = p.s.ox; st[ST_OMEGA + 1] = p.s.oy; st[ST_OMEGA + 2] = p.s.oz;
    for (int i = 0; i < 3; ++i) {
        st[ST_FDIST + i] = p.f[i];
        st[ST_POSTAR + i] = p.ptar[i];
        st[ST_VELTAR + i] = p.vtar[i];
        st[ST_ACCTAR + i] = p.atar[i];
    }
    st[ST_TIME] = __int_as_float(p.time);
    for (int i = ST_TIME + 1; i < COVO_STATE_FLOATS; ++i) st[i] = 0.0f;
}

// pid.py:38-76 (geometric PD position + P attitude; the "angle" handed to Rodrigues' formula is |e3 x z_d|, as there)
__device__ __forceinline__ void pid_action(const PidState &p, const PidArgs &A, float (&act)[4])
{
    const float x = p.s.qx, y = p.s.qy, z = p.s.qz, w = p.s.qw;
    const float Q[3][3] = {{w * w + x * x - y * y - z * z, 2 * (x * y - z * w), 2 * (x * z + y * w)},
                           {2 * (x * y + z * w), w * w - x * x + y * y - z * z, 2 * (y * z - x * w)},
                           {2 * (x * z - y * w), 2 * (y * z + x * w), w * w - x * x - y * y + z * z}};
    const float pos[3] = {p.s.px, p.s.py, p.s.pz}, vel[3] = {p.s.vx, p.s.vy, p.s.vz};
    float fd[3];
    for (int i = 0; i < 3; ++i)
        fd[i] = A.pid_m * ((i == 2 ? A.pid_g : 0.0f) - A.Kp * (pos[i] - p.ptar[i]) - A.Kd * (vel[i] - p.vtar[i]) + p.atar[i]);
    float thrust = Q[0][2] * fd[0] + Q[1][2] * fd[1] + Q[2][2] * fd[2];  // (Q^T f_d)[2]
    thrust = fminf(fmaxf(thrust, 0.0f), A.pid_max_thrust);
    float n = sqrtf(fd[0] * fd[0] + fd[1] * fd[1] + fd[2] * fd[2]);
    n = n < 1e-3f ? 1e-3f : n;
    const float zd[3] = {fd[0] / n, fd[1] / n, fd[2] / n};
    const float aa[3] = {-zd[1], zd[0], 0.0f};  // e3 x z_d
    float angle = sqrtf(aa[0] * aa[0] + aa[1] * aa[1]);
    angle = angle < 1e-3f ? 5e-4f : angle;
    float ax[3] = {0.0f, 0.0f, 1.0f};
    if (!(angle < 1e-3f)) { ax[0] = aa[0] / angle; ax[1] = aa[1] / angle; ax[2] = aa[2] / angle; }
    const float an = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);  // geom.py:108
    ax[0] /= an; ax[1] /= an; ax[2] /= an;
    const float K[3][3] = {{0.0f, -ax[2], ax[1]}, {ax[2], 0.0f, -ax[0]}, {-ax[1], ax[0], 0.0f}};
    const float sn = sinf(angle), cs = 1.0f - cosf(angle);
    float Rd[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float kk = 0.0f;
            for (int l = 0; l < 3; ++l) kk += K[i][l] * K[l][j];
            Rd[i][j] = (i == j ? 1.0f : 0.0f) + sn * K[i][j] + cs * kk;
        }
    float Re[3][3];  // R_d^T Q
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float acc = 0.0f;
            for (int l = 0; l < 3; ++l) acc += Rd[l][i] * Q[l][j];
            Re[i][j] = acc;
        }
    const float err[3] = {Re[2][1] - Re[1][2], Re[0][2] - Re[2][0], Re[1][0] - Re[0][1]};  // vee(R_e - R_e^T)
    act[0] = thrust / A.pid_max_thrust * 2.0f - 1.0f;
    for (int i = 0; i < 3; ++i) act[1 + i] = -A.Kp_att * err[i] / A.pid_max_omega[i];
}

// the disturbance of the NEXT step from the PRE-step state (free.py:147); sk = the key this step_env call receives
__device__ __forceinline__ void pid_next_force(const PidState &p, const PidArgs &A, const uint32_t (&sk)[2], bool deterministic,
                                               float (&fn)[3])
{
    const dm::Model &m = A.dist;
    const bool draws = (m.kind == COVO_DISTURB_GAUSSIAN && !deterministic) ||
                       ((m.kind == COVO_DISTURB_PERIODIC || m.kind == COVO_DISTURB_MIXED) && (p.time % m.period) == 0);
    uint32_t dk[2] = {0u, 0u};
    if (draws) dm::disturb_key(sk, dk);
    const float vel[3] = {p.s.vx, p.s.vy, p.s.vz};
    for (int i = 0; i < 3; ++i) fn[i] = dm::next_force(m, dk, p.time, vel[i], p.f[i], i, deterministic);
}

// one env step with action `act` (quadrotor.py:215-263, free.py:114-155); f_next = the disturbance of the NEXT step
__device__ __forceinline__ void pid_env_step(PidState &p, const float (&act)[4], const PidArgs &A, const float (&f_next)[3])
{
    const float a0 = qm::clip11_(qm::clip11_(act[0])), a1 = qm::clip11_(qm::clip11_(act[1]));
    const float a2 = qm::clip11_(qm::clip11_(act[2])), a3 = qm::clip11_(qm::clip11_(act[3]));
    qm::dyn_step<float, float>(p.s, a0, a1, a2, a3, A.c, p.f[0], p.f[1], p.f[2]);
    p.time += 1;
    const int idx = p.time < 0 ? 0 : (p.time > A.T - 1 ? A.T - 1 : p.time);
    for (int i = 0; i < 3; ++i) {
        p.f[i] = f_next[i];
        p.ptar[i] = A.pos_traj[3 * idx + i];
        p.vtar[i] = A.vel_traj[3 * idx + i];
        p.atar[i] = A.acc_traj[3 * idx + i];
    }
}

// covo.py:80-90: the chain of start states (one lane; lanes 0..2 of the wave form the three components of a step's next force)
__global__ __launch_bounds__(64) void pid_chain_kernel(const PidArgs A)
{
    __shared__ float zf[3];
    const int lane = threadIdx.x;
    PidState p;
    pid_load(p, A.state0);
    uint32_t key[2] = {A.key[0], A.key[1]};
    for (int t = 0; t < A.n_steps; ++t) {
        if (lane == 0) {
            pid_store(p, A.states + (size_t)t * COVO_STATE_FLOATS);
            if (A.keys) { A.keys[2 * t] = key[0]; A.keys[2 * t + 1] = key[1]; }
        }
        // rng_step, key = split(key) (consumed by the PID call); rng_step, key = split(key) (the env step's key)
        uint32_t k1[2], rs[2], k2[2];
        pid_split(key, 1u, k1);
        pid_split(k1, 0u, rs);
        pid_split(k1, 1u, k2);
        key[0] = k2[0];
        key[1] = k2[1];
        // step_env(rs, deterministic=False): the next force from the PRE-step state
        if (lane < 3) {
            const dm::Model &m = A.dist;
            const bool draws = m.kind == COVO_DISTURB_GAUSSIAN ||
                               ((m.kind == COVO_DISTURB_PERIODIC || m.kind == COVO_DISTURB_MIXED) && (p.time % m.period) == 0);
            uint32_t dk[2] = {0u, 0u};
            if (draws) dm::disturb_key(rs, dk);
            const float vel = lane == 0 ? p.s.vx : (lane == 1 ? p.s.vy : p.s.vz);
            zf[lane] = dm::next_force(m, dk, p.time, vel, p.f[lane], lane, false);
        }
        __syncthreads();
        float act[4];
        pid_action(p, A, act);
        const float fn[3] = {zf[0], zf[1], zf[2]};
        pid_env_step(p, act, A, fn);
        __syncthreads();
    }
}

// covo.py:58-76: H deterministic PID steps from every start state; their (unclipped) actions are the nominal mean
__global__ __launch_bounds__(64) void pid_nominal_kernel(const PidArgs A)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= A.n_steps) return;
    PidState p;
    pid_load(p, A.states + (size_t)t * COVO_STATE_FLOATS);
    // the scan's carry key at this start state (covo.py:73-75): only the drawing models thread it
    const bool keyed = A.keys != nullptr && (A.dist.kind == COVO_DISTURB_PERIODIC || A.dist.kind == COVO_DISTURB_MIXED);
    uint32_t key[2] = {keyed ? A.keys[2 * t] : 0u, keyed ? A.keys[2 * t + 1] : 0u};
    for (int k = 0; k < COVO_H; ++k) {
        float act[4];
        pid_action(p, A, act);
        for (int d = 0; d < 4; ++d) A.a_means[(size_t)t * COVO_NA + 4 * k + d] = act[d];
        // mppi_rollout_fn (covo.py:58-70): rng_act, key = split(key); rng_step, key = split(key); step_env(rng_step, deterministic=True)
        uint32_t rs[2] = {0u, 0u};
        if (keyed) {
            uint32_t k1[2], k2[2];
            pid_split(key, 1u, k1);
            pid_split(k1, 0u, rs);
            pid_split(k1, 1u, k2);
            key[0] = k2[0];
            key[1] = k2[1];
        }
        float fn[3];
        pid_next_force(p, A, rs, true, fn);  // deterministic=True: only the gaussian model is off (quadrotor.py:234-235)
        pid_env_step(p, act, A, fn);
    }
}

int launch_pid_nominal(const float *state0, const float *pos_traj, const float *vel_traj, const float *acc_traj, int T,
                       const covo_env_params &p, const covo_env_params &pid_params, float Kp, float Kd, float Kp_att,
                       uint32_t key0, uint32_t key1, int n_steps, float *states, float *a_means, uint32_t *keys_out,
                       hipStream_t s)
{
    PidArgs A;
    A.state0 = state0;
    A.pos_traj = pos_traj;
    A.vel_traj = vel_traj;
    A.acc_traj = acc_traj;
    A.states = states;
    A.a_means = a_means;
    A.T = T;
    A.n_steps = n_steps;
    A.key[0] = key0;
    A.key[1] = key1;
    A.c = make_consts<float>(p);
    A.pid_m = pid_params.m;
    A.pid_g = pid_params.g;
    A.pid_max_thrust = pid_params.max_thrust;
    for (int i = 0; i < 3; ++i) A.pid_max_omega[i] = pid_params.max_omega[i];
    A.Kp = Kp;
    A.Kd = Kd;
    A.Kp_att = Kp_att;
    A.dist = dm::make_model(p);
    A.keys = keys_out;
    if ((A.dist.kind == COVO_DISTURB_PERIODIC || A.dist.kind == COVO_DISTURB_MIXED) && keys_out == nullptr) {
        covo_set_error("pid_nominal: disturb_kind=%d draws in the nominal rollouts: keys_out must be given", A.dist.kind);
        return COVO_E_BADARG;
    }
    hipLaunchKernelGGL(pid_chain_kernel, dim3(1), dim3(64), 0, s, A);
    hipLaunchKernelGGL(pid_nominal_kernel, dim3((n_steps + 63) / 64), dim3(64), 0, s, A);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}
