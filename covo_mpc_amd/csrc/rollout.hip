// rollout.hip -- the fused N-sample x H-step rollout kernel (gfx950).
//
// Replaces lax.scan(H) of vmap(N) Quad3D.step_env + done-freeze + discounted cost:
//   quadjax/controllers/covo.py:227-263, mppi.py:71-106  (scan, freeze, cost)
//   quadjax/envs/quadrotor.py:215-263, 479-490           (step_env, raw_step, is_terminal)
//   quadjax/dynamics/free.py:74-155, utils.py:266-294    (dynamics, reward)  -> quad_model.hpp
//
// The only per-sample HBM traffic is the stripe-ordered action read (32 x 16 B, each wave-load one contiguous 1 KiB)
// and the 4-byte cost write: 516 B / sample.  HBM-roofline kernel by bytes, VALU-issue bound in fact
// (~100-135 instructions per sample-step), so instruction count and waves per SIMD matter as much as bytes.
//   rollout_pipe3_kernel (rollout_pipe.hpp): the product kernel -- three waves per 64 samples (attitude / translation /
//     reward) pipelined through
//     LDS rings, with or without the position statistics of covo.py:281 (STATS) and the softmax records (REC);
//   rollout_kernel (below, ROLLOUT_LAB_BASELINE only): round 1's one-lane-per-sample kernel, the baseline of
//     scripts/probe/rollout_lab.hip -- not part of the library.
#include "rollout_launch.hpp"
#include "disturb_model.hpp"

// sums the per-block position statistics in fp64: out[k*6+i]
// one workgroup per (step, statistic) column: 256 threads stride over the per-block partials, fixed-order tree
// (a single 192-thread workgroup walking all blocks serially took 77 us at N = 65 536 and 1.4 ms at N = 1 048 576)
__global__ __launch_bounds__(256) void pos_stats_finalize_kernel(const double *__restrict__ ws, int nblocks,
                                                                 double *__restrict__ out)
{
    __shared__ double red[256];
    const int i = blockIdx.x, tid = threadIdx.x;
    double acc = 0.0;
    for (int b = tid; b < nblocks; b += 256) acc += ws[(size_t)b * (COVO_H * 6) + i];
    red[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) out[i] = red[0];
}

#ifdef ROLLOUT_LAB_BASELINE
template <bool STATS, bool DISC1, bool CLIP, bool BATCHED = false>
static void launch_rollout_pf(const RolloutArgs &A, const RolloutArgs *batch, int nbatch, int grid, bool deep, hipStream_t s)
{
    if (deep)
        hipLaunchKernelGGL((rollout_kernel<STATS, DISC1, CLIP, COVO_H, BATCHED>), dim3(grid, nbatch), dim3(RO_BLOCK), 0, s, A, batch);
    else
        hipLaunchKernelGGL((rollout_kernel<STATS, DISC1, CLIP, 8, BATCHED>), dim3(grid, nbatch), dim3(RO_BLOCK), 0, s, A, batch);
}
#endif

void fill_rollout_args(RolloutArgs &A, const float *state, const float *pos_traj, const float *vel_traj, int T,
                       const covo_env_params &p, const float *f_shared, const float *a, int N, float discount,
                       float *cost, float *groupmin, double *stats_ws, const float *f_shared_dev, const float *f_tab,
                       int xcd_groups, int nbatch)
{
    if (xcd_groups <= 0) xcd_groups = noise_gemm_groups_per_workgroup(N, nbatch);  // the producer is the noise GEMM unless told otherwise
    A.state = state;
    A.pos_traj = pos_traj;
    A.vel_traj = vel_traj;
    A.a = reinterpret_cast<const float4 *>(a);
    A.cost = cost;
    A.groupmin = groupmin;
    A.stats_ws = stats_ws;
    A.N = N;
    A.T = T;
    A.max_steps = p.max_steps_in_episode;
    A.discount = discount;
    for (int i = 0; i < 3; ++i) A.f_shared[i] = f_shared ? f_shared[i] : 0.0f;
    A.f_shared_dev = f_shared_dev;
    // the producers run one 32-sample tile per wave up to N = 65 536: workgroup g of the noise GEMM (noise_gemm_block_threads / 128
    // groups of 64 samples; MPPI's block-diagonal kernel: 256 samples = 4 groups per workgroup row) sits on XCD g % 8, and the
    // rollout's workgroups take the groups their own XCD's L2 has just been written with (a mismatch costs MPPI's rollout 11 us)
    A.xcd_remap = (N % 2048 == 0 && N / 128 <= 512) ? xcd_groups : 0;
    A.records = nullptr;
    A.inv_lam = 0.0f;
    A.merge_ticket = nullptr;  // (step_small.hip sets these for its own launch)
    A.merge_out = nullptr;
    A.merge_mean_old = nullptr;
    A.merge_gamma = 1.0f;
    A.merge_final = 0;
    A.clip = 1;
    A.rollover = p.rollover_terminate != 0;
    A.reward = p.reward_kind;
    // which disturbance variant (rollout_pipe.hpp): none / gaussian -> one shared vector; periodic / sin -> the per-step
    // table; drag / mixed -> the per-sample force
    const dm::Model m = dm::make_model(p);
    A.fdist = (m.kind == COVO_DISTURB_PERIODIC || m.kind == COVO_DISTURB_SIN) ? 1
              : (m.kind == COVO_DISTURB_DRAG || m.kind == COVO_DISTURB_MIXED) ? 2 : 0;
    A.f_tab = reinterpret_cast<const float4 *>(f_tab);
    A.drag_k = dm::drag_coeff(m);
    for (int i = 0; i < 3; ++i) A.drag_off[i] = 0.5f * m.dp[i];
    A.c = make_consts<float>(p);
}

// nbatch == 0: one rollout described by A; else nbatch instances described by the device array `batch` (all with A's
// N, discount, clip, rollover, reward and disturbance contract -- A is only used to pick the kernel variant, the same one a
// single launch would get)
template <bool BATCHED>
static int dispatch_rollout(const RolloutArgs &A, const RolloutArgs *batch, int nbatch, double *pos_stats, hipStream_t s)
{
    const int N = A.N;
    const int nb = BATCHED ? nbatch : 1;
    const bool stats = !BATCHED && pos_stats != nullptr;
    const int groups = pipe_groups(N, nb);
    if (A.reward == COVO_REWARD_PENYAW && A.fdist == 0) {
        if (A.discount == 1.0f) launch_pipe3_family<true, BATCHED, 0, 0>(A, batch, nb, groups, stats, s);
        else launch_pipe3_family<false, BATCHED, 0, 0>(A, batch, nb, groups, stats, s);
    } else if (A.reward == COVO_REWARD_PENYAW) {
        launch_rollout_variant_r0(A, batch, nb, BATCHED, groups, stats, s);
    } else {
        launch_rollout_variant_r1(A, batch, nb, BATCHED, groups, stats, s);
    }
    if (stats) {
        // position statistics: the same pipelined kernel, its T waves also summing the new positions (rollout_pipe.hpp);
        // one fp64 partial per workgroup and (step, statistic), summed by one workgroup per column
        const int grid = ((N + COVO_WAVE - 1) / COVO_WAVE + groups - 1) / groups;
        hipLaunchKernelGGL(pos_stats_finalize_kernel, dim3(COVO_H * 6), dim3(256), 0, s, A.stats_ws, grid, pos_stats);
    }
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_rollout(const float *state, const float *pos_traj, const float *vel_traj, int T, const covo_env_params &p,
                   const float *f_shared, const float *a, int N, float discount, bool trust_clipped, float *cost,
                   float *groupmin, double *pos_stats, double *stats_ws, hipStream_t s, const float *f_shared_dev,
                   float *records, float lam, const float *f_tab, int xcd_groups, bool propagate_nan)
{
    RolloutArgs A;
    fill_rollout_args(A, state, pos_traj, vel_traj, T, p, f_shared, a, N, discount, cost, groupmin, stats_ws, f_shared_dev, f_tab,
                      xcd_groups);
    if (A.fdist != 0 && f_tab == nullptr) {
        covo_set_error("rollout: disturb_kind=%d needs the per-step disturbance table (covo_disturb_table)", p.disturb_kind);
        return COVO_E_BADARG;
    }
    if (p.reward_kind != COVO_REWARD_PENYAW && p.reward_kind != COVO_REWARD_REALWORLD) {
        covo_set_error("rollout: reward_kind=%d", p.reward_kind);
        return COVO_E_BADARG;
    }
    A.clip = trust_clipped ? 0 : (propagate_nan ? 2 : 1);
    A.records = records;
    A.inv_lam = records ? 1.0f / lam : 0.0f;
    return dispatch_rollout<false>(A, nullptr, 0, pos_stats, s);
}

// workgroups (= softmax records, rollout_record) a rollout over N samples (x nbatch instances) is launched with, per
// instance -- mirrors dispatch_rollout
int rollout_workgroups(int N, bool stats, int nbatch)
{
    (void)stats;  // the statistics ride in the same launch shape
    const int ng = (N + COVO_WAVE - 1) / COVO_WAVE, groups = pipe_groups(N, nbatch);
    return (ng + groups - 1) / groups;
}

// ---- env-batched rollout: one launch, workgroup row y = instance y (step.hip: covo_mpc_step_batched)
size_t rollout_args_bytes(int n) { return (size_t)n * sizeof(RolloutArgs); }

void rollout_fill_args(void *out, int index, const float *state, const float *pos_traj, const float *vel_traj, int T,
                       const covo_env_params &p, const float *a, int N, float discount, float *cost, float *groupmin,
                       const float *f_shared_dev, float *records, float lam, bool trust_clipped, const float *f_tab)
{
    RolloutArgs &A = reinterpret_cast<RolloutArgs *>(out)[index];
    // f_tab: this instance's rows of the step's disturbance tables (periodic / sin / drag / mixed; all instances share the kind)
    fill_rollout_args(A, state, pos_traj, vel_traj, T, p, nullptr, a, N, discount, cost, groupmin, nullptr, f_shared_dev, f_tab, 0, 2);
    A.clip = trust_clipped ? 0 : 1;
    A.records = records;
    A.inv_lam = records ? 1.0f / lam : 0.0f;
}

// every instance must share instance 0's N, discount, clip, rollover, reward and disturbance kind (checked by the caller)
int launch_rollout_batched(const void *args_host, const void *args_dev, int nbatch, hipStream_t s)
{
    return dispatch_rollout<true>(reinterpret_cast<const RolloutArgs *>(args_host)[0],
                                  reinterpret_cast<const RolloutArgs *>(args_dev), nbatch, nullptr, s);
}
