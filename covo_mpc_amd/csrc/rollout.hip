// rollout.hip -- the fused N-sample x H-step rollout kernel (gfx950).
//
// Replaces lax.scan(H) of vmap(N) Quad3D.step_env + done-freeze + discounted cost:
//   quadjax/controllers/covo.py:227-263, mppi.py:71-106  (scan, freeze, cost)
//   quadjax/envs/quadrotor.py:215-263, 479-490           (step_env, raw_step, is_terminal)
//   quadjax/dynamics/free.py:74-155, utils.py:266-294    (dynamics, reward)  -> quad_model.hpp
//
// One lane = one sample.  The 13-float state, the frozen reward and the running cost stay in
// VGPRs for the whole horizon; wave-uniform data (targets, discount, time-out flag) sit in a
// 1 KiB LDS window read by broadcast; the only per-sample HBM traffic is the stripe-ordered
// action read (32 x 16 B, each wave-load one contiguous 1 KiB) and the 4-byte cost write:
// 516 B / sample.  HBM-roofline kernel; ~165 VALU slots per step put it on the
// fp32 ridge, so instruction count matters as much as bytes.
#include "covo_common.hpp"

struct RolloutArgs {
    const float *state;
    const float *pos_traj;
    const float *vel_traj;
    const float4 *a;   // [H][N]
    float *cost;       // [N]
    float *blockmin;   // [gridDim.x] or null
    float *stats_ws;   // [gridDim.x][H*6] or null
    int N, T, max_steps;
    float discount;
    float f_shared[3];
    qm::Consts<float> c;
};

constexpr int RO_BLOCK = 256;
constexpr int RO_PF = 4;  // action prefetch distance (steps)

template <bool STATS>
__global__ __launch_bounds__(RO_BLOCK) void rollout_kernel(const RolloutArgs A)
{
    __shared__ float4 win[COVO_H][2];  // {ptar.xyz, vtar.x}, {vtar.yz, discount^k, timed_out}
    __shared__ float red[RO_BLOCK / COVO_WAVE];
    __shared__ float sacc[STATS ? COVO_H * 6 : 1];

    const int tid = threadIdx.x;
    const float *__restrict__ st = A.state;
    const int time0 = __float_as_int(st[ST_TIME]);
    if (tid < COVO_H) {
        const int k = tid;
        float px, py, pz, vx, vy, vz;
        if (k == 0) {  // step 0 uses the state's own targets
            px = st[ST_POSTAR + 0]; py = st[ST_POSTAR + 1]; pz = st[ST_POSTAR + 2];
            vx = st[ST_VELTAR + 0]; vy = st[ST_VELTAR + 1]; vz = st[ST_VELTAR + 2];
        } else {       // free.py:150-155: targets = traj[time+1] after each step; gather clamps
            int idx = time0 + k;
            idx = idx < 0 ? 0 : (idx > A.T - 1 ? A.T - 1 : idx);
            px = A.pos_traj[3 * idx + 0]; py = A.pos_traj[3 * idx + 1]; pz = A.pos_traj[3 * idx + 2];
            vx = A.vel_traj[3 * idx + 0]; vy = A.vel_traj[3 * idx + 1]; vz = A.vel_traj[3 * idx + 2];
        }
        float disc = 1.0f;
        for (int i = 0; i < k; ++i) disc *= A.discount;  // discount^k (covo.py:258)
        win[k][0] = make_float4(px, py, pz, vx);
        win[k][1] = make_float4(vy, vz, disc, (time0 + k >= A.max_steps) ? 1.0f : 0.0f);
    }
    if (STATS) {
        for (int i = tid; i < COVO_H * 6; i += RO_BLOCK) sacc[i] = 0.0f;
    }
    __syncthreads();

    const int n_raw = blockIdx.x * RO_BLOCK + tid;
    const bool valid = n_raw < A.N;
    const int n = valid ? n_raw : A.N - 1;
    const float4 *__restrict__ ap = A.a + n;
    const size_t stride = (size_t)A.N;

    qm::State<float> s;
    s.px = st[ST_POS + 0]; s.py = st[ST_POS + 1]; s.pz = st[ST_POS + 2];
    s.vx = st[ST_VEL + 0]; s.vy = st[ST_VEL + 1]; s.vz = st[ST_VEL + 2];
    s.qx = st[ST_QUAT + 0]; s.qy = st[ST_QUAT + 1]; s.qz = st[ST_QUAT + 2]; s.qw = st[ST_QUAT + 3];
    s.ox = st[ST_OMEGA + 0]; s.oy = st[ST_OMEGA + 1]; s.oz = st[ST_OMEGA + 2];
    const float p0x = s.px, p0y = s.py, p0z = s.pz;
    const float f0x = st[ST_FDIST + 0], f0y = st[ST_FDIST + 1], f0z = st[ST_FDIST + 2];
    const qm::Consts<float> c = A.c;

    float4 ring[RO_PF];
#pragma unroll
    for (int i = 0; i < RO_PF; ++i) ring[i] = ap[(size_t)i * stride];

    float acc = 0.0f, r_before = 0.0f;   // covo.py:246-247
    bool done_before = false;

#pragma unroll
    for (int k = 0; k < COVO_H; ++k) {
        const float4 w0 = win[k][0], w1 = win[k][1];
        // reward / termination of the PRE-step state (quadrotor.py:243-244)
        float r = qm::reward<float, float>(s, w0.x, w0.y, w0.z, w0.w, w1.x, w1.y);
        const float pmax = fmaxf(fmaxf(fabsf(s.px), fabsf(s.py)), fabsf(s.pz));
        const bool done = (w1.w != 0.0f) | (pmax > c.pos_limit);
        r = done_before ? r_before : r;  // covo.py:233
        done_before = done_before | done;
        r_before = r;
        acc = fmaf(w1.z, r, acc);        // covo.py:257-261

        const float4 av = ring[k % RO_PF];
        if (k + RO_PF < COVO_H) ring[k % RO_PF] = ap[(size_t)(k + RO_PF) * stride];
        const float fx = (k == 0) ? f0x : A.f_shared[0];
        const float fy = (k == 0) ? f0y : A.f_shared[1];
        const float fz = (k == 0) ? f0z : A.f_shared[2];
        // step_env / raw_step clip (quadrotor.py:223,258) -- idempotent on already-clipped stripes
        qm::dyn_step<float, float>(s, qm::clip11_(av.x), qm::clip11_(av.y), qm::clip11_(av.z), qm::clip11_(av.w), c,
                                   fx, fy, fz);
        if (STATS) {  // covo.py:234-237: post-step positions, shifted by the initial position
            const float dx = valid ? s.px - p0x : 0.0f, dy = valid ? s.py - p0y : 0.0f, dz = valid ? s.pz - p0z : 0.0f;
            const float v0 = wave_sum(dx), v1 = wave_sum(dy), v2 = wave_sum(dz);
            const float v3 = wave_sum(dx * dx), v4 = wave_sum(dy * dy), v5 = wave_sum(dz * dz);
            if ((tid & (COVO_WAVE - 1)) == 0) {
                atomicAdd(&sacc[k * 6 + 0], v0); atomicAdd(&sacc[k * 6 + 1], v1); atomicAdd(&sacc[k * 6 + 2], v2);
                atomicAdd(&sacc[k * 6 + 3], v3); atomicAdd(&sacc[k * 6 + 4], v4); atomicAdd(&sacc[k * 6 + 5], v5);
            }
        }
    }
    const float cost = -acc;  // covo.py:263
    if (valid) A.cost[n] = cost;

    if (A.blockmin != nullptr) {
        const float wm = wave_min(valid ? cost : __builtin_inff());
        if ((tid & (COVO_WAVE - 1)) == 0) red[tid / COVO_WAVE] = wm;
        __syncthreads();
        if (tid == 0) {
            float m = red[0];
#pragma unroll
            for (int i = 1; i < RO_BLOCK / COVO_WAVE; ++i) m = fminf(m, red[i]);
            A.blockmin[blockIdx.x] = m;
        }
    }
    if (STATS) {
        __syncthreads();
        for (int i = tid; i < COVO_H * 6; i += RO_BLOCK) A.stats_ws[(size_t)blockIdx.x * (COVO_H * 6) + i] = sacc[i];
    }
}

// sums the per-block position statistics in fp64: out[k*6+i]
__global__ __launch_bounds__(256) void pos_stats_finalize_kernel(const float *__restrict__ ws, int nblocks,
                                                                 double *__restrict__ out)
{
    const int i = threadIdx.x;
    if (i >= COVO_H * 6) return;
    double acc = 0.0;
    for (int b = 0; b < nblocks; ++b) acc += (double)ws[(size_t)b * (COVO_H * 6) + i];
    out[i] = acc;
}

int launch_rollout(const float *state, const float *pos_traj, const float *vel_traj, int T, const covo_env_params &p,
                   const float *f_shared, const float *a, int N, float discount, float *cost, float *blockmin,
                   double *pos_stats, float *stats_ws, hipStream_t s)
{
    RolloutArgs A;
    A.state = state;
    A.pos_traj = pos_traj;
    A.vel_traj = vel_traj;
    A.a = reinterpret_cast<const float4 *>(a);
    A.cost = cost;
    A.blockmin = blockmin;
    A.stats_ws = stats_ws;
    A.N = N;
    A.T = T;
    A.max_steps = p.max_steps_in_episode;
    A.discount = discount;
    for (int i = 0; i < 3; ++i) A.f_shared[i] = f_shared ? f_shared[i] : 0.0f;
    A.c = make_consts<float>(p);
    const int grid = (N + RO_BLOCK - 1) / RO_BLOCK;
    if (pos_stats != nullptr) {
        hipLaunchKernelGGL(rollout_kernel<true>, dim3(grid), dim3(RO_BLOCK), 0, s, A);
        hipLaunchKernelGGL(pos_stats_finalize_kernel, dim3(1), dim3(256), 0, s, stats_ws, grid, pos_stats);
    } else {
        hipLaunchKernelGGL(rollout_kernel<false>, dim3(grid), dim3(RO_BLOCK), 0, s, A);
    }
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}
