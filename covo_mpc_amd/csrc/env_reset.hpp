// env_reset.hpp -- reset_env on the device (gfx950): what BaseEnvironment.step selects when an episode is done.
//
// quadjax/envs/base.py:22-40: key, key_reset = split(key); step_env(key, ...); obs_re, info_re, state_re = reset_env(key_reset,
// params); every leaf of (state, info, obs) = select(done, reset, stepped).  eval_env steps through exactly that
// (quadjax/envs/quadrotor.py:531-538): an episode that leaves the 3 m box (or rolls over) continues from a FRESH state on a FRESH
// reference trajectory, with the controller's state carried on.  reset_env (quadrotor.py:363-370) = get_zero_state (265-312):
//   traj_key, disturb_key, key = split(key_reset, 3); pos/vel/acc_traj = generate_traj(traj_key); zero state, quat (0,0,0,1),
//   targets = row 0, time 0, f_disturb = uniform(disturb_key, (3,), -disturb_scale, disturb_scale)
// then info_key, key = split(key_reset); get_info(info_key, state, state): the noisy copy from split(info_key, 5)[0..3].
//
// The generators follow covo_mpc_amd/dynamics/utils.py (the host mirror of quadjax/dynamics/utils.py:49-53, 87-130, 133-180,
// 183-251; numpy fp64, rounded to fp32 at the end) operation by operation with fp contraction off, on the Philox draws of
// covo_mpc_amd/random.py.  sin / cos / acos / atan2 are ocml's fp64 functions here and libm's there: both are accurate to an ulp
// or two of fp64, i.e. the fp32 roundings agree except when an fp64 value sits within ~1e-16 of a rounding boundary (about one
// element in 10^7: tests/test_gpu_reset.py compares the generated trajectories with the host's and allows one fp32 ulp).
// One wave per env instance; runs only on the step at which an instance terminates.
#pragma once
#include "covo_common.hpp"
#include "disturb_model.hpp"

namespace er {

// element i of uniform(key, (n,), lo, hi, dtype=float64): random.py:72-77 (block i / 4 of the 0xB175 stream, word i % 4)
__device__ inline double uniform64(const uint32_t (&key)[2], int i, double lo, double hi)
{
#pragma clang fp contract(off)
    uint32_t b[4];
    rngd::philox4x32_10((uint32_t)(i >> 2), 0u, 0u, 0xB175u, key[0], key[1], b);
    const double u = ((double)(b[i & 3] >> 8) + 0.5) / 16777216.0;
    return lo + (hi - lo) * u;
}

// row r of the three trajectories; row 0 (the reset state's targets, quadrotor.py:285-287) also goes to row0[9] in LDS
__device__ inline void store_row(float *pos_traj, float *vel_traj, float *acc_traj, float *row0, int r, const double (&p)[3],
                                 const double (&v)[3], const double (&a)[3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        pos_traj[3 * r + i] = (float)p[i];
        vel_traj[3 * r + i] = (float)v[i];
        acc_traj[3 * r + i] = (float)a[i];
        if (r == 0) {
            row0[i] = (float)p[i];
            row0[3 + i] = (float)v[i];
            row0[6 + i] = (float)a[i];
        }
    }
}

// utils.py:87-130 (f1 = .2, f2 = .4) and 133-180 (f1 = f2 = .1): rows lane, lane + 64, ... of the T = max_steps + 50 rows
__device__ inline void lissa_rows(const uint32_t (&key)[2], double f1, double f2, int T, double dt, float *pos_traj, float *vel_traj,
                                  float *acc_traj, float *row0, int lane)
{
#pragma clang fp contract(off)
    uint32_t key_amp[2], key_phase[2];
    dm::split(key, 0u, key_amp);
    dm::split(key, 1u, key_phase);
    const double pi = 3.141592653589793;
    const double w1 = 2 * pi * f1, w2 = 2 * pi * f2;
    double amp[3][2], phase[3][2], p0[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            amp[i][j] = uniform64(key_amp, 2 * i + j, -1.0, 1.0);
            phase[i][j] = uniform64(key_phase, 2 * i + j, -pi, pi);
        }
        p0[i] = amp[i][0] * sin(w1 * 0.0 + phase[i][0]) + amp[i][1] * sin(w2 * 0.0 + phase[i][1]);
    }
    for (int r = lane; r < T; r += 64) {
        const double ts = (double)r * dt;
        double p[3], v[3], a[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double x1 = w1 * ts + phase[i][0], x2 = w2 * ts + phase[i][1];
            const double s1 = sin(x1), s2 = sin(x2), c1 = cos(x1), c2 = cos(x2);
            p[i] = (amp[i][0] * s1 + amp[i][1] * s2) - p0[i];
            v[i] = amp[i][0] * w1 * c1 + amp[i][1] * w2 * c2;
            a[i] = -amp[i][0] * (w1 * w1) * s1 - amp[i][1] * (w2 * w2) * s2;
        }
        store_row(pos_traj, vel_traj, acc_traj, row0, r, p, v, a);
    }
}

// utils.py:183-251 incl. its quirks (shared key arrays :187-188, segments 0 and 1 both on keys[1] :238-241, distance U(1, 1.5)
// :219, velocity / (point_per_seg + 1) :231-236).  kp: LDS, (num_seg + 1) key points x 3, filled by lane 0.
__device__ inline void zigzag_rows(const uint32_t (&key)[2], int max_steps, int T, double dt, float *pos_traj, float *vel_traj,
                                   float *acc_traj, float *row0, int lane, double (*kp)[3])
{
#pragma clang fp contract(off)
    const int pps = 40;
    const int num_seg = max_steps / pps + 1;  // rows = num_seg * 40 = T
    if (lane == 0) {
        const double pi = 3.141592653589793;
        uint32_t k[2];
        dm::split(key, 0u, k);
        double prev[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) prev[i] = uniform64(k, i, -1.0, 1.0);
        {
            const double nrm = sqrt(prev[0] * prev[0] + prev[1] * prev[1] + prev[2] * prev[2]);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                prev[i] = prev[i] / nrm * 0.1;
                kp[0][i] = prev[i];
            }
        }
        dm::split(key, 1u, k);
        for (int s = 0; s < num_seg; ++s) {
            const double nrm = sqrt(prev[0] * prev[0] + prev[1] * prev[1] + prev[2] * prev[2]);
            const double tc0 = -prev[0] / nrm, tc1 = -prev[1] / nrm, tc2 = -prev[2] / nrm;
            const double d_theta = uniform64(k, 0, -pi / 3, pi / 3), d_phi = uniform64(k, 1, -pi / 3, pi / 3);
            const double theta = acos(tc2) + d_theta;
            const double phi = atan2(tc1, tc0) + d_phi;
            const double dir[3] = {sin(theta) * cos(phi), sin(theta) * sin(phi), cos(theta)};
            const double distance = 1.0 + 0.5 * uniform64(k, 0, 0.0, 1.0);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                prev[i] = prev[i] + distance * dir[i];
                kp[s + 1][i] = prev[i];
            }
            dm::split(key, (uint32_t)(s + 1), k);  // k = keys[i + 1]: segment 1 runs on keys[1] again
        }
    }
    __syncthreads();
    for (int r = lane; r < T; r += 64) {
        const int s = r / pps, j = r % pps;
        double p[3], v[3];
        const double a[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double delta = kp[s + 1][i] - kp[s][i];
            const double step = delta / (double)pps;              // np.linspace(prev, next, 40, endpoint=False)
            p[i] = ((double)j * step + kp[s][i]) - kp[0][i];      // ... - pos[0]
            v[i] = delta / (double)(pps + 1) * 1.0 / dt;
        }
        store_row(pos_traj, vel_traj, acc_traj, row0, r, p, v, a);
    }
}

// rows a generator of kind `kind` produces for an episode of max_steps steps (the T the caller's buffers must have)
__host__ __device__ inline int traj_rows(int kind, int max_steps)
{
    switch (kind) {
    case COVO_TRAJ_FIXED: return max_steps;
    case COVO_TRAJ_LISSA:
    case COVO_TRAJ_LISSA_SLOW: return max_steps + 50;
    case COVO_TRAJ_ZIGZAG: return (max_steps / 40 + 1) * 40;
    default: return -1;
    }
}

struct ResetArgs {
    int kind, max_steps, T;
    double dt, disturb_scale;
};

// One wave.  Writes the new trajectories (global memory), the reset state into sst[0..32) and the 13 observation-noise normals of
// reset_env's get_info into z[3..16); row0[9] and kp[][3] are LDS scratch.  Every lane must call it (barriers inside); the caller barriers before reading sst / z.
__device__ __noinline__ void reset_env(const ResetArgs &R, const uint32_t (&step_key)[2], float *pos_traj, float *vel_traj,
                                       float *acc_traj, float *sst, float *z, float *row0, double (*kp)[3])
{
    const int lane = threadIdx.x;
    uint32_t key_reset[2], traj_key[2], disturb_key[2];
    dm::split(step_key, 1u, key_reset);    // base.py:22: key, key_reset = split(key)
    dm::split(key_reset, 0u, traj_key);    // quadrotor.py:267: traj_key, disturb_key, key = split(key, 3)
    dm::split(key_reset, 1u, disturb_key);
    switch (R.kind) {
    case COVO_TRAJ_LISSA: lissa_rows(traj_key, 0.2, 0.4, R.T, R.dt, pos_traj, vel_traj, acc_traj, row0, lane); break;
    case COVO_TRAJ_LISSA_SLOW: lissa_rows(traj_key, 0.1, 0.1, R.T, R.dt, pos_traj, vel_traj, acc_traj, row0, lane); break;
    case COVO_TRAJ_ZIGZAG: zigzag_rows(traj_key, R.max_steps, R.T, R.dt, pos_traj, vel_traj, acc_traj, row0, lane, kp); break;
    default:
        for (int r = lane; r < 3 * R.T; r += 64) pos_traj[r] = vel_traj[r] = acc_traj[r] = 0.0f;
        if (lane < 9) row0[lane] = 0.0f;
        break;
    }
    // the observation noise of reset_env's get_info (quadrotor.py:365-366, 322-350): info_key = split(key_reset)[0]
    if (lane >= 3 && lane < 16) {
        const int grp = lane < 6 ? 1 : lane < 9 ? 2 : lane < 13 ? 3 : 4;
        const int base = grp == 1 ? 3 : grp == 2 ? 6 : grp == 3 ? 9 : 13;
        uint32_t info_key[2], nk[2];
        dm::split(key_reset, 0u, info_key);
        dm::split(info_key, (uint32_t)(grp - 1), nk);
        // element lane - base of normal(nk, (n,)): random.py:79-95 (the same formula as env_step.hip's host_normal)
        const int n = grp == 3 ? 4 : 3, i = lane - base;
        uint32_t b1[4], b2[4];
        rngd::philox4x32_10((uint32_t)(i >> 2), 0u, 0u, 0xB175u, nk[0], nk[1], b1);
        rngd::philox4x32_10((uint32_t)((n + i) >> 2), 0u, 0u, 0xB175u, nk[0], nk[1], b2);
        const double u1 = ((double)(b1[i & 3] >> 8) + 0.5) / 16777216.0;
        const double u2 = ((double)(b2[(n + i) & 3] >> 8) + 0.5) / 16777216.0;
        z[lane] = (float)(sqrt(-2.0 * log(u1)) * cos(2.0 * 3.141592653589793 * u2));
    }
    __syncthreads();  // row0 is in LDS
    if (lane < COVO_STATE_FLOATS) {
        float v = 0.0f;
        if (lane == ST_QUAT + 3) v = 1.0f;
        else if (lane >= ST_FDIST && lane < ST_FDIST + 3)
            v = (float)uniform64(disturb_key, lane - ST_FDIST, -R.disturb_scale, R.disturb_scale);  // quadrotor.py:300-305
        else if (lane >= ST_POSTAR && lane < ST_POSTAR + 9) v = row0[lane - ST_POSTAR];  // pos_tar, vel_tar, acc_tar are contiguous
        else if (lane == ST_TIME) v = __int_as_float(0);
        sst[lane] = v;
    }
}

}  // namespace er
