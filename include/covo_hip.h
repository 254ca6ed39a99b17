/* covo_hip.h -- C ABI of libcovo_hip.so: the MI355X (gfx950) sampling-MPC inner loop.
 *
 * The reference (LeCAR-Lab/CoVO-MPC, package quadjax) has NO plugin/FFI boundary for
 * this path: it is pure Python-on-JAX and the boundary is the Python controller
 * protocol quadjax/controllers/base.py:5-19.  This header is therefore the
 * build-defined C ABI that covo_mpc_amd's Python controllers (the mirror of
 * quadjax.controllers) bind with ctypes; each entry point cites the reference
 * lines whose XLA-lowered ops it replaces.  INTEGRATION.md shows the ctypes stub
 * a quadjax maintainer would add.
 *
 * Conventions
 *   - extern "C", every function returns int: 0 = ok, otherwise a COVO_E_* code
 *     (negative) or a hipError_t (positive); covo_last_error() gives the text.
 *   - All tensor arguments are CALLER-OWNED DEVICE pointers (torch data_ptr()),
 *     fp32, contiguous, unless marked [host].  The library owns only the opaque
 *     handle and a small device workspace allocated in covo_create().
 *   - Every launch goes to the caller's hipStream_t (passed as void*); there are
 *     no hidden synchronisations or allocations on these paths (graph-capturable).
 *   - A handle is single-threaded and bound to the device current at creation.
 *
 * Data layouts in HBM
 *   state    float[COVO_STATE_FLOATS]: the (noisy) env state the controller plans
 *            from (quadjax/controllers/covo.py:198), packed as
 *            pos[0:3] vel[3:6] quat_xyzw[6:10] omega[10:13] f_disturb[13:16]
 *            pos_tar[16:19] vel_tar[19:22] acc_tar[22:25] time(int32 bits)[25]
 *   traj     pos_traj / vel_traj: float[T][3] reference trajectory (EnvState3D)
 *   eps      float[N][n]      standard-normal draws, sample-major, n = H*du = 128
 *   a        float[H][N][4]   clipped sampled actions in "stripe" order: one
 *            float4 (thrust, wx, wy, wz) per (step, sample); a wave of the rollout
 *            kernel reads one contiguous 1 KiB stripe per step.
 *   cost     float[N]
 *   partial  float[COVO_PARTIAL_FLOATS] = {m, s, v[128], pad}: online-softmax
 *            record of one shard (m = min cost, s = sum w, v = sum w*a)
 */
#ifndef COVO_HIP_H
#define COVO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COVO_ABI_VERSION 7

#define COVO_H 32            /* horizon (compile-time in the fused kernels)          */
#define COVO_DU 4            /* action dim, quadjax/envs/quadrotor.py:198            */
#define COVO_NA (COVO_H * COVO_DU) /* n = 128 flattened actions, index 4*t + d        */
#define COVO_STATE_FLOATS 32
#define COVO_PARTIAL_FLOATS 132
#define COVO_POS_STATS_DOUBLES (COVO_H * 6) /* per step: sum(pos-c)[3], sum((pos-c)^2)[3] */

#define COVO_FLAG_ACTIONS_CLIPPED 1 /* covo_config.flags: every `a` handed to covo_rollout_cost is already
                                       clipped to [-1,1] (true for covo_noise_* outputs): skip step_env's re-clip */

#define COVO_FLAG_NO_GRAPH 2         /* covo_config.flags: covo_mpc_step always launches eagerly (no hipGraph) */
#define COVO_FLAG_SHARED_DEVICE 4    /* covo_config.flags: other processes / streams compete for this GPU: no launch may depend on
                                        its workgroups being co-resident (the Sigma chain then runs every phase as its own
                                        launch instead of folding them into two persistent launches).  Set it as well when TWO
                                        HANDLES of one process run covo-online / covo_sigma steps concurrently on different
                                        streams: two persistent launches side by side can hold the workgroup slots each other's
                                        missing workgroups need (their barriers then time out: COVO_DEVSTAT_GRID_BARRIER).  Steps on
                                        one stream, and any number of handles used one after the other, need nothing. */

#define COVO_FLAG_PROPAGATE_NAN 8    /* covo_config.flags: the action clips (covo.py:224, mppi.py:66, quadrotor.py:223,258) keep a NaN
                                        like jnp.clip = minimum(maximum(x, lo), hi) does.  Default (flag off): the kernels clip with
                                        v_med3 / maxNum / minNum, where a NaN operand loses -- a NaN sample becomes -1.
                                        The one place quadjax produces NaNs by itself on this path: covo-offline on `hovering`, whose
                                        Sigma-table row 0 is NaN (norm'(0)); there quadjax's mean is NaN from the first step on, and
                                        with this flag so is this library's; without it the episode carries on. */

#define COVO_MODE_MPPI 0
#define COVO_MODE_COVO_ONLINE 1
#define COVO_MODE_COVO_OFFLINE 2

#define COVO_E_BADARG (-1)
#define COVO_E_NOHANDLE (-2)
#define COVO_E_UNSUPPORTED (-3)
#define COVO_E_DEVICE (-4)           /* a kernel of an EARLIER call on this handle reported a failure (covo_device_status) */

#define COVO_RANK_RECORD_FLOATS (COVO_PARTIAL_FLOATS + 2 * COVO_POS_STATS_DOUBLES) /* 516: the record one rank of a sample-sharded
                                        step contributes to the exchange: {m, s, v[128], pad[2]} + the 192 fp64 position sums of
                                        covo.py:281 (zeros when not requested): ONE message of 2 064 bytes per rank and step */
#define COVO_COV_FLOATS (COVO_H * 10)  /* MPPI's covariance adaptation (mppi.py:119-125): the 10 second moments i <= j of the 4 action
                                        components of every step, sum_n w_n d_i d_j about the shifted OLD mean */
#define COVO_RANK_RECORD_COV_FLOATS (COVO_PARTIAL_FLOATS + COVO_COV_FLOATS + 2 * COVO_POS_STATS_DOUBLES) /* 836: the rank record of a
                                        sample-sharded MPPI step with gamma_sigma != 0: {m, s, v[128], pad[2]} + 320 second moments +
                                        the 192 fp64 position sums -- still ONE message (3 344 bytes) per rank and step */
#define COVO_EXCHANGE_HANDLE_BYTES 128 /* opaque inter-process handle of a rank's exchange buffer (covo_exchange_create): the hipIpc
                                        handle + whether the buffer is fine-grained + the PCI bus id of its device */

#define COVO_DEVSTAT_EXCHANGE 2      /* covo_exchange_records: a peer's record did not arrive within the exchange's time-out
                                        (covo_exchange_set_timeout, default 60 s); the gathered records are NaN */
#define COVO_DEVSTAT_ADJOINT 4       /* covo_hessian / covo_mpc_step: a hyper-dual workgroup of the adjoint Hessian's chain launch
                                        gave up waiting for a costate row of the same launch (0.2 s: the costate workgroup was not
                                        co-resident); that call's Hessian, Sigma and L are NaN */
#define COVO_DEVSTAT_GRID_BARRIER 1  /* a grid barrier of the Sigma chain's persistent launches timed out (its workgroups were not
                                        co-resident within 0.2 s: GPU shared with other work); that call's Sigma / L are NaN */

/* covo_env_params.reward_kind: which reward Quad3D binds to env.reward_fn (quadjax/envs/quadrotor.py:49-84) */
#define COVO_REWARD_PENYAW 0     /* tracking_penyaw_reward_fn (dynamics/utils.py:285-294): tasks tracking, tracking_zigzag, hovering */
#define COVO_REWARD_REALWORLD 1  /* tracking_realworld_reward_fn (utils.py:297-313): task tracking_slow */

/* covo_env_params.disturb_kind: Quad3D(disturb_type=...) -> get_quadrotor_1st_order_dyn (dynamics/free.py:8-72) */
#define COVO_DISTURB_NONE 0      /* free.py:72 */
#define COVO_DISTURB_GAUSSIAN 1  /* free.py:66-70: dyn_noise_scale * normal(key, (3,)); 0 under deterministic=True (quadrotor.py:234) */
#define COVO_DISTURB_PERIODIC 2  /* free.py:10-24: redrawn uniform(-scale, scale) when time % period == 0, held otherwise */
#define COVO_DISTURB_SIN 3       /* free.py:27-38: a function of time and disturb_params */
#define COVO_DISTURB_DRAG 4      /* free.py:41-47: -|scale| rel |rel| / 1.5^2, rel = vel - disturb_params[:3] / 2 (per sample) */
#define COVO_DISTURB_MIXED 5     /* free.py:50-56: (drag + sin + periodic) / 3 */

/* covo_env_params.reset_traj: the trajectory generator Quad3D binds to self.generate_traj (quadjax/envs/quadrotor.py:49-84), i.e.
 * what reset_env (quadrotor.py:265-312, 363-370) draws when BaseEnvironment.step auto-resets a finished episode
 * (quadjax/envs/base.py:22-40).  COVO_TRAJ_NONE switches the device env's auto-reset off (done is logged, the state flies on). */
#define COVO_TRAJ_NONE 0
#define COVO_TRAJ_FIXED 1        /* generate_fixed_traj (dynamics/utils.py:49-53): task hovering; T = max_steps_in_episode rows of 0 */
#define COVO_TRAJ_LISSA 2        /* generate_lissa_traj (utils.py:87-130): task tracking; T = max_steps_in_episode + 50 */
#define COVO_TRAJ_LISSA_SLOW 3   /* generate_lissa_traj_slow (utils.py:133-180): task tracking_slow; T = max_steps_in_episode + 50 */
#define COVO_TRAJ_ZIGZAG 4       /* generate_zigzag_traj (utils.py:183-251): task tracking_zigzag; T = (max_steps_in_episode / 40 + 1) * 40 */

/* covo_disturb_table key threading (who calls step_env with which key) */
#define COVO_DISTURB_KEYS_SHARED 0    /* every step uses the SAME step key: the controllers' rollouts (covo.py:225,231; mppi.py:69,74) */
#define COVO_DISTURB_KEYS_HESSIAN 1   /* per step rng_k, key = split(key); step_env(rng_k): get_hessian (covo.py:150-153) */
#define COVO_DISTURB_KEYS_NOMINAL 2   /* per step _, key = split(key); rng_step, key = split(key): covo-offline's nominal rollout (covo.py:58-70) */

typedef struct covo_ctx *covo_handle_t;

/* Rollout-relevant subset of EnvParams3D (quadjax/dynamics/dataclass.py:40-100). [host] */
typedef struct covo_env_params {
    float max_thrust;       /* .8   */
    float max_torque[3];    /* 9e-3, 9e-3, 2e-3 (cancels: quadrotor.py:260 * free.py:122) */
    float max_omega[3];     /* 10, 10, 3 */
    float dt;               /* .02  */
    float g;                /* 9.81 */
    float m;                /* .027 */
    float action_scale;     /* 1    */
    float alpha_bodyrate;   /* .5   */
    int32_t max_steps_in_episode; /* 300 */
    float pos_limit;        /* 3.0, quadrotor.py:484 */
    int32_t rollover_terminate; /* 1: is_terminal also fires on quat[3] < cos(pi/4) or any |omega| > 100 (quadrotor.py:486-490),
                                 *    i.e. Quad3D(disable_rollover_terminate=False), the constructor's default; quadjax's main()
                                 *    builds its env with disable_rollover_terminate=True (quadrotor.py:779) -> 0 */
    int32_t reward_kind;        /* COVO_REWARD_*: env.reward_fn (quadrotor.py:56,66,73,82) */
    int32_t disturb_kind;       /* COVO_DISTURB_*: Quad3D(disturb_type=...) (quadrotor.py:35,87-89) */
    int32_t disturb_period;     /* 50   (dataclass.py:86) */
    float disturb_scale;        /* .2   (dataclass.py:87) */
    float disturb_params[6];    /* 0    (dataclass.py:88; domain randomisation / reset draw them, quadrotor.py:151,171) */
    float dyn_noise_scale;      /* .05  (dataclass.py:93): scale of the gaussian model (zeroed by deterministic=True) */
    /* ---- auto-reset of the device env (covo_env_step*, covo_run_episode*; base.py:22-40).  Ignored by every other entry point. */
    int32_t reset_traj;         /* COVO_TRAJ_*; COVO_TRAJ_NONE (0) = no auto-reset */
    int32_t reserved0;          /* 0 */
    double reset_dt;            /* dt as the trajectory generators receive it (quadrotor.py:52,60,68,77: the host's double .02) */
    double reset_disturb_scale; /* f_disturb ~ U(-s, s) at reset (quadrotor.py:300-305): the host's double .2 */
} covo_env_params;

typedef struct covo_config {
    int32_t n_local;        /* samples handled by this handle (this rank's shard)  */
    int32_t H;              /* must equal COVO_H                                   */
    int32_t du;             /* must equal COVO_DU                                  */
    float lam;              /* temperature lambda, controllers/covo.py:266         */
    float discount;         /* controllers/covo.py:258                             */
    int32_t flags;          /* COVO_FLAG_* bits                                    */
} covo_config;

const char *covo_last_error(void);
int covo_abi_version(void);

/* Replaces nothing in the reference: lifetime of the opaque handle + workspace. */
int covo_create(const covo_config *cfg, covo_handle_t *out);
int covo_destroy(covo_handle_t h);

/* Sticky device-side status of the handle: COVO_DEVSTAT_* bits raised by kernels of earlier (asynchronous) calls, read from
 * host-mapped memory without synchronising.  While it is non-zero every compute entry point returns COVO_E_DEVICE instead of
 * enqueueing more work on poisoned data; clear != 0 resets it (after the caller has dealt with the failed step, e.g. by
 * re-creating the handle with COVO_FLAG_SHARED_DEVICE).  Returns the status before clearing, or COVO_E_NOHANDLE. */
int covo_device_status(covo_handle_t h, int32_t clear);

/* Counter-based N(0,1) fill (Philox4x32-10 + Box-Muller) keyed by
 * (key0, key1, global sample id, column): results do not depend on how samples are
 * sharded.  Stands in for jax.random.split + normal inside
 * jax.random.multivariate_normal (controllers/covo.py:212-220, mppi.py:53-65);
 * jax's threefry bitstream is unpinned/unavailable, so the stream is build-defined.
 * eps_out: float[n_samples][n_cols], rows = global ids sample_offset .. +n_samples. */
int covo_randn(covo_handle_t h, uint32_t key0, uint32_t key1, int64_t sample_offset, int32_t n_samples,
               int32_t n_cols, float *eps_out, void *stream);

/* The same draws from jax.random's OWN bitstream (threefry2x32-20, jax 0.4.x default layout), so that a
 * JAX-equipped machine can replay a run without shipping epsilon:  row n of eps_out (float[n_samples][128]) is
 *   mppi == 0:  jax.random.normal(jax.random.split(act_key, n_total)[sample_offset + n], (128,))        (covo.py:213-220)
 *   mppi != 0:  concat_t jax.random.normal(jax.random.split(jax.random.split(act_key, n_total)[..], H)[t], (4,))  (mppi.py:53-60)
 * with act_key = (key0, key1).  Twin of covo_mpc_amd/random_jax.py, which is pinned to the Random123 threefry vectors and to
 * the values jax's documentation prints; ~2x the cost of covo_randn, meant for replay / parity runs (the kernel-by-kernel
 * path), not for the fused step. */
int covo_randn_jax(covo_handle_t h, uint32_t key0, uint32_t key1, int64_t n_total, int64_t sample_offset, int32_t n_samples,
                   int32_t mppi, float *eps_out, void *stream);

/* a = clip(mu + L eps, -1, 1): jax.random.multivariate_normal's `mean + factor @ eps`
 * followed by jnp.clip (controllers/covo.py:215-224).  fp32 MFMA GEMM
 * (v_mfma_f32_32x32x2_f32; each dot product is an ascending-k fmaf chain, bit-exact).
 * L: float[128][128] row-major lower-triangular Cholesky factor of a_cov (entries
 * above the diagonal are ignored); mu: float[128]; eps: float[N][128]; a_out: float[H][N][4]. */
int covo_noise_gemm(covo_handle_t h, const float *L, const float *mu, const float *eps, int32_t N, float *a_out,
                    void *stream);

/* Same, with epsilon drawn inside the kernel (never written to HBM): row n of the virtual eps matrix is
 * the covo_randn row of global sample id sample_offset + n for the same key -- bit-identical to
 * covo_randn followed by covo_noise_gemm. */
int covo_noise_gemm_philox(covo_handle_t h, const float *L, const float *mu, uint32_t key0, uint32_t key1,
                           int64_t sample_offset, int32_t N, float *a_out, void *stream);

/* MPPI's per-step sampling (controllers/mppi.py:53-66): a[t] = clip(mu[t] + Ls[t] eps[t]).
 * Ls: float[H][4][4] lower factors; eps: float[N][H][4]; a_out: float[H][N][4]. */
int covo_noise_blockdiag(covo_handle_t h, const float *Ls, const float *mu, const float *eps, int32_t N,
                         float *a_out, void *stream);

int covo_noise_blockdiag_philox(covo_handle_t h, const float *Ls, const float *mu, uint32_t key0, uint32_t key1,
                                int64_t sample_offset, int32_t N, float *a_out, void *stream);

/* The fused N x H rollout: lax.scan(H) of vmap(N) Quad3D.step_env + done-freeze +
 * discounted cost (controllers/covo.py:227-263, mppi.py:71-106; envs/quadrotor.py:215-263,
 * 479-490; dynamics/free.py:74-155; dynamics/utils.py:266-294).  Per-sample state lives in
 * registers / LDS; only cost[N] (+ one min per 64-sample wave) is written.  Actions are re-clipped like
 * step_env does unless the handle was created with COVO_FLAG_ACTIONS_CLIPPED.  Termination follows
 * params->rollover_terminate (quadrotor.py:479-490).
 * The disturbance acting during rollout step 0 is the state's own f_disturb; for steps k >= 1 (free.py:147) it follows
 * params->disturb_kind:
 *   NONE / GAUSSIAN: f_disturb_shared [host float[3]]: the single vector every sample and step receives from the shared
 *     step_key (0 for CoVO's deterministic=True; dyn_noise_scale * normal for MPPI); NULL = 0.
 *   PERIODIC / SIN / DRAG / MIXED: f_disturb_steps [DEVICE float[COVO_H][4]], the table covo_disturb_table builds
 *     (COVO_DISTURB_KEYS_SHARED): row k = {g_k[3], c_k}, f_k = c_drag * drag(vel_{k-1}) + c_k * f_{k-1} + g_k with
 *     c_drag = 1 (DRAG), 1/3 (MIXED), 0 otherwise -- PERIODIC and SIN are wave-uniform per step (c_k = 0, g_k = the force
 *     itself: no per-sample work), DRAG and MIXED keep a per-sample force.  f_disturb_shared is then ignored.
 * pos_stats (nullable): double[COVO_H*6] accumulators, zeroed by the call, receiving
 *   per-step sum(pos - pos0) and sum((pos - pos0)^2) over samples of the post-step
 *   positions (controllers/covo.py:234-237,281); pos0 = state pos.
 * groupmin (nullable): float[ceil(N/64)] minimum of cost over each group of 64 consecutive samples. */
int covo_rollout_cost(covo_handle_t h, const float *state, const float *pos_traj, const float *vel_traj, int32_t T,
                      const covo_env_params *params, const float *f_disturb_shared, const float *f_disturb_steps,
                      const float *a, int32_t N, float *cost_out, float *groupmin, double *pos_stats, void *stream);

/* The per-step disturbance table of a rollout that starts at `state` (DEVICE float[batch][COVO_STATE_FLOATS]; it supplies
 * time and, for PERIODIC's hold, f_disturb): out [DEVICE float[batch][COVO_H][4]], rows as described at covo_rollout_cost,
 * for params->disturb_kind = PERIODIC / SIN / DRAG / MIXED (NONE: zeros; GAUSSIAN: g_k = dyn_noise_scale * normal unless
 * `deterministic`).  The uniform draws of PERIODIC / MIXED (free.py:16-21) come from the key step_env would hand to
 * disturb_func -- disturb_key = split(split(split(k)[1])[0])[0] for step_env(k) (quadrotor.py:262, free.py:136,144) -- with k
 * threaded as `key_mode` (COVO_DISTURB_KEYS_*) says from the batch entry's key: keys_dev [DEVICE uint32[batch][2]] or, when
 * NULL, (key0, key1) for every entry.  Philox keys / host formulas of covo_mpc_amd/random.py (the stream is build-defined). */
int covo_disturb_table(covo_handle_t h, const covo_env_params *params, const float *state, int32_t batch,
                       const uint32_t *keys_dev, uint32_t key0, uint32_t key1, int32_t key_mode, int32_t deterministic,
                       float *out, void *stream);

/* controllers/covo.py:281 (mppi.py:134) from the sums covo_rollout_cost / covo_mpc_step leave in pos_stats (after the
 * all-reduce when the samples are sharded): pos_mean[k][i] = pos0[i] + S1/n, pos_std[k][i] = sqrt(max(S2/n - (S1/n)^2, 0))
 * (ddof 0), both float[COVO_H][3]; n_total = the global sample count.  One launch instead of the caller's elementwise ops. */
int covo_pos_info(covo_handle_t h, const double *pos_stats, const float *state, int64_t n_total, float *pos_mean_out,
                  float *pos_std_out, void *stream);

/* softmax weights + weighted sum as ONE online-softmax record of this shard
 * (controllers/covo.py:266-272 before normalisation): two-stage wavefront reduction.
 * groupmin (nullable): the per-64-sample minima covo_rollout_cost left for exactly this
 *   cost/N (saves one pass over cost); recomputed internally when null.
 * partial_out: float[COVO_PARTIAL_FLOATS]. */
int covo_softmax_reduce(covo_handle_t h, const float *cost, const float *a, int32_t N, const float *groupmin,
                        float *partial_out, void *stream);

/* Single-shard finish: covo_softmax_reduce + covo_merge(G=1) without materialising the record:
 * a_mean_out = gamma_mean * sum_n w_n a_n + (1-gamma_mean) * a_mean_old (covo.py:266-275). */
int covo_softmax_update(covo_handle_t h, const float *cost, const float *a, int32_t N, const float *groupmin,
                        const float *a_mean_old, float gamma_mean, float *a_mean_out, void *stream);

/* MPPI with covariance adaptation (controllers/mppi.py:109-125, gamma_sigma != 0; quadjax's own factory fixes gamma_sigma = 0,
 * envs/quadrotor.py:715): the update of covo_softmax_update plus
 *   a_cov_out[t] = gamma_sigma sum_n w_n (a_n[t] - a_mean_out[t]) (a_n[t] - a_mean_out[t])^T + (1 - gamma_sigma) a_cov_old[t]
 * with the NEW mean, as the reference does.  a_cov_*: float[H][4][4] (a_cov_old already shifted, mppi.py:43-49); in place allowed.
 * The weighted second moments are accumulated about a_mean_old (the mean the samples were drawn around).  Single shard. */
int covo_softmax_update_cov(covo_handle_t h, const float *cost, const float *a, int32_t N, const float *groupmin,
                            const float *a_mean_old, float gamma_mean, const float *a_cov_old, float gamma_sigma,
                            float *a_mean_out, float *a_cov_out, void *stream);

/* The same on sample-sharded ranks (round 4): covo_softmax_reduce_cov leaves this rank's UNNORMALISED record
 * {m, s, v[128], pad[2], S2[320]} (S2 = the weighted second moments about a_mean_old, which every rank shares) in the first
 * COVO_PARTIAL_FLOATS + COVO_COV_FLOATS floats of record_out -- a COVO_RANK_RECORD_COV_FLOATS rank record whose last 192 doubles are
 * the position sums (or zeros); after ONE exchange of the G records (all-gather, or covo_exchange_records_cov)
 * covo_merge_ranks_cov forms the new mean and the adapted covariances identically on every rank and, if asked, adds the position sums. */
int covo_softmax_reduce_cov(covo_handle_t h, const float *cost, const float *a, int32_t N, const float *groupmin,
                            const float *a_mean_old, float *record_out, void *stream);
int covo_merge_ranks_cov(covo_handle_t h, const float *records /* [G][COVO_RANK_RECORD_COV_FLOATS] */, int32_t G,
                         const float *a_mean_old, float gamma_mean, const float *a_cov_old, float gamma_sigma, float *a_mean_out,
                         float *a_cov_out, double *pos_stats_out, void *stream);

/* Merge G shard records (this GPU's, or the all-gathered records of all ranks), normalise,
 * blend with the old mean (controllers/covo.py:270-275):
 *   a_mean_out = gamma_mean * (sum_g v_g e^{-(m_g-m)/lam}) / (sum_g s_g e^{-(m_g-m)/lam})
 *              + (1-gamma_mean) * a_mean_old.         a_mean_*: float[128] (index 4t+d). */
int covo_merge(covo_handle_t h, const float *partials, int32_t G, const float *a_mean_old, float gamma_mean,
               float *a_mean_out, void *stream);

/* Sample-sharded step: merge the all-gathered RANK records (float[G][COVO_RANK_RECORD_FLOATS]: rank g's
 * covo_mpc_step wrote its {m, s, v} to partial_out = record and its position sums to pos_stats = record + COVO_PARTIAL_FLOATS)
 * like covo_merge, and (pos_stats_out != NULL) sum the ranks' position sums into pos_stats_out [double[COVO_POS_STATS_DOUBLES]]
 * for covo_pos_info -- the statistics travel in the same message as the softmax partial, not in a second collective. */
int covo_merge_ranks(covo_handle_t h, const float *records, int32_t G, const float *a_mean_old, float gamma_mean,
                     float *a_mean_out, double *pos_stats_out, void *stream);

/* covo_merge_ranks over all-gathered records of `record_floats` floats each: COVO_RANK_RECORD_FLOATS, or
 * COVO_RANK_RECORD_COV_FLOATS -- the records of a core built for MPPI's covariance adaptation that runs a step with
 * gamma_sigma == 0: {m, s, v[128], pad[2]} in front, the 320 second-moment slots unused, the position sums at the END of the
 * wider record.  Any other record_floats is refused. */
int covo_merge_ranks_wide(covo_handle_t h, const float *records, int32_t G, int32_t record_floats, const float *a_mean_old,
                          float gamma_mean, float *a_mean_out, double *pos_stats_out, void *stream);

/* The exchange of the rank records as direct peer writes instead of a collective (exchange.hip).  Setup, once:
 * every rank calls covo_exchange_create (allocates its buffer, returns its inter-process handle), the G handles are
 * all-gathered by the caller (any transport: they are 64 opaque bytes), every rank calls covo_exchange_connect with all G of
 * them (rank order).  Per step: covo_exchange_records enqueues, on `stream`, the push of this rank's record into every peer's
 * buffer and the wait for all G records, which land in gathered_out [float[G][COVO_RANK_RECORD_FLOATS], device]; no host
 * synchronisation.  A peer that does not deliver within the time-out (covo_exchange_set_timeout; default 60 s, or
 * COVO_EXCHANGE_TIMEOUT_S at covo_exchange_create) raises COVO_DEVSTAT_EXCHANGE and leaves NaN records.  With a
 * connected exchange covo_run_episode also runs on sample-sharded handles (args->partial_out = this rank's record).
 * covo_exchange_connect returns COVO_E_UNSUPPORTED, before mapping anything, when two ranks sit on DIFFERENT devices and either
 * buffer is coarse-grained (the fine-grained allocation failed): remote xGMI writes into it would not be guaranteed visible.
 * NOT measured over xGMI (the build pool has single-GPU boxes): the host mirror keeps the collective unless asked. */
int covo_exchange_create(covo_handle_t h, int32_t world, int32_t rank, void *handle_out /* [host COVO_EXCHANGE_HANDLE_BYTES] */);
int covo_exchange_connect(covo_handle_t h, const void *handles /* [host world x COVO_EXCHANGE_HANDLE_BYTES] */);
int covo_exchange_set_timeout(covo_handle_t h, double seconds);
int covo_exchange_records(covo_handle_t h, const float *record, float *gathered_out, void *stream);
/* the same for the COVO_RANK_RECORD_COV_FLOATS record kind (gathered_out: float[G][COVO_RANK_RECORD_COV_FLOATS]) */
int covo_exchange_records_cov(covo_handle_t h, const float *record, float *gathered_out, void *stream);

/* PCI bus id ("0000:c1:00.0") of HIP device `device` of this process -> out (NUL-terminated, len >= 16): the PHYSICAL identity
 * of a GPU.  Device indices are per-process (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES renumber them: every isolated rank sees
 * index 0), so ranks that must know whether they share a GPU compare this, not the index. */
int covo_device_bus_id(int32_t device, char *out, int32_t len);

/* a_mean <- [a_mean[1:], a_mean[-1]] (controllers/covo.py:201-203).  in != out. */
int covo_shift_mean(covo_handle_t h, const float *a_mean_in, float *a_mean_out, void *stream);

/* Exact Hessian of the CoVO objective -(sum_k r(s_k) + r(s_0)) w.r.t. the flattened mean
 * actions: jax.jacfwd(jax.jacfwd(get_cumulated_cost)) of controllers/covo.py:134-185
 * (deterministic, no discount, no done-freeze), fp64.  Second-order adjoint: one hyper-dual evaluation of
 * the step model per (time, pair of step inputs), a costate and a sensitivity recursion, and
 * sum_k S_k^T M_k S_k on the matrix cores (hessian_adj.hip).  May (re)allocate scratch, with a stream sync,
 * the first time a batch size is seen.  R_out: double[batch][128][128]; state/a_mean are strided by
 * COVO_STATE_FLOATS / 128 per batch entry; traj is shared.  The reward is params->reward_kind's.
 * f_disturb_steps [DEVICE float[batch][COVO_H][4], nullable]: covo_disturb_table(COVO_DISTURB_KEYS_HESSIAN, deterministic = 1)
 * -- required for params->disturb_kind PERIODIC / SIN / DRAG / MIXED (get_hessian's deterministic=True only switches the
 * gaussian model off, quadrotor.py:234-235); NULL = no force after step 0.  DRAG / MIXED make the force part of the
 * differentiated state: those two run the kernels' 16-component instantiation (state + force; hessian_adj.hip adj16). */
int covo_hessian(covo_handle_t h, const float *state, const float *pos_traj, const float *vel_traj, int32_t T,
                 const covo_env_params *params, const float *a_mean, const float *f_disturb_steps, int32_t batch,
                 double *R_out, void *stream);

/* The same Hessian by one hyper-dual ROLLOUT per unordered pair of actions (hessian.hip): ~4x slower,
 * derived independently; kept as the cross-check of covo_hessian. */
int covo_hessian_pairs(covo_handle_t h, const float *state, const float *pos_traj, const float *vel_traj, int32_t T,
                       const covo_env_params *params, const float *a_mean, const float *f_disturb_steps, int32_t batch,
                       double *R_out, void *stream);

/* CoVO's optimal covariance (controllers/covo.py:116-132) and the lower Cholesky factor
 * jax.random.multivariate_normal takes of it (covo.py:216-218).  The reference's
 *   eigh -> shift spectrum to min 1e-2 -> U diag(exp(log_s)) U^T (det = sigma^(2n)) -> symmetrise
 * is the matrix function Sigma = c (R + delta I)^(-1/2), delta = 1e-2 - lambda_min(R); it is evaluated
 * WITHOUT an eigendecomposition (fp64 MFMA GEMM pipeline: repeated squaring + Rayleigh-Ritz for
 * lambda_min, coupled Newton-Schulz for the inverse square root, Cholesky for log det; sigma_ns.hip)
 * and agrees with LAPACK eigh to ~1e-15.  Sigma is rounded to fp32 (the reference's a_cov dtype)
 * before its Cholesky factor is taken.  May (re)allocate scratch, with a stream sync, the first time a
 * batch size is seen.
 * R: double[batch][128][128]; Sigma_out: float[batch][128][128] (nullable);
 * L_out: float[batch][128][128] lower-triangular. */
int covo_sigma(covo_handle_t h, const double *R, int32_t batch, float sample_sigma, float *Sigma_out,
               float *L_out, void *stream);

/* Same result by an explicit symmetric eigendecomposition (one-sided cyclic block Jacobi, fp64, one
 * workgroup per matrix; sigma.hip).  Slower (latency of 10-12 sweeps); kept as an independent check. */
int covo_sigma_jacobi(covo_handle_t h, const double *R, int32_t batch, float sample_sigma, float *Sigma_out,
                      float *L_out, void *stream);

/* ---- Experiment switches of ONE handle (round 6; rounds 1-5 kept them process-global).  A new handle takes its defaults from the
 * environment (COVO_FUSE_SMALL, COVO_STREAM_GEMM, COVO_FOLD_BEGIN, COVO_NS_DEFLATE, COVO_NS_RITZ_INSIDE; csrc/step.hip:
 * covo_default_opts); the setters below change them for THAT handle only and make it re-capture its step graphs at the next step.
 * They exist for A/B measurements and the parity tests: every pair of settings gives the same bits unless stated otherwise.
 * A handle is single-threaded (as every entry point that takes one). */

/* 0: covo-offline / MPPI steps of <= 256 sample groups run their staged launches (begin | noise | rollout + records | merge)
 * instead of the ONE fused launch of csrc/step_small.hip (the default where eligible). */
int covo_debug_set_fuse_small(covo_handle_t h, int on);

/* 0: covo-online's noise GEMM runs as a launch of its own behind the Sigma chain (rounds 1-4) instead of streamed under the
 * factorisation inside the chain's finalize launch (the default for one matrix on a GPU of one's own).  Same actions, a_cov and
 * means bit for bit. */
int covo_debug_set_stream_gemm(covo_handle_t h, int on);

/* 0: eager covo-online steps keep the begin launch (mean shift, key derivation, sequence bump) instead of folding its work into the
 * Hessian's first launch (the default; every launch of a folded step reads args->state where it lies instead of the handle's
 * fixed-address copy).  Same bits (tests/test_gpu_parity.py::test_folded_begin_equals_the_begin_launch). */
int covo_debug_set_fold_begin(covo_handle_t h, int on);

/* Debug aid: copy `count` doubles from offset `offset_doubles` of the Sigma pipeline's scratch (layout in
 * sigma_ns.hip: 11 matrices [batch][128][128], then 3 712 doubles of slots per matrix -- SC_* in sigma_ns.hip: lambda_min, delta,
 * scale, the iterate the result was taken from, iteration counts, barrier status ... --, then the filter's iterate buffers) to
 * `out` (device or pinned host). */
int covo_debug_sigma_workspace(covo_handle_t h, double *out, int64_t offset_doubles, int64_t count, void *stream);
/* How many of the eigh-free Sigma chain's last Chebyshev squarings / Newton-Schulz iterations run inside the two persistent
 * launches (phases separated by barriers inside the launch) instead of as one / two launches each.  Defaults: one matrix (15, 11)
 * = all but the first of each; batched launches (15, 4).  The call sets both; (64, 64) = all but the first of each, (0, 0) = every
 * phase its own launch, (-1, -1) = back to the defaults.  The result does not depend on it bit for bit. */
int covo_debug_set_ns_tail(covo_handle_t h, int n_squarings, int n_iters);
/* 0 switches the deflation of the bottom eigenpair in the eigh-free Sigma chain off (sigma_ns.hip: the Newton-Schulz iteration then
 * runs on B itself, ~2 iterations more); default on.  Results agree to fp64 rounding either way (not bit for bit: another
 * iteration sequence). */
int covo_debug_set_ns_deflate(covo_handle_t h, int on);
/* 0: the one-matrix Sigma chain runs its two persistent launches (squarings with the evaluations inside | Newton-Schulz iterations)
 * as TWO launches (rounds 4-5) instead of one whose iteration workgroups wait inside for the chain's result (csrc/sigma_ns.hip:
 * ns_chain_kernel; the default for covo-online steps on a GPU of one's own; COVO_NS_MERGED=0 in the environment).  Same Sigma and L
 * bit for bit. */
int covo_debug_set_ns_merged(covo_handle_t h, int on);
/* 1 makes the Sigma chain's persistent launches behave as if their workgroups had NOT all landed on one XCD (sigma_ns.hip: every
 * access stays an agent-scope atomic, COH_AGENT) -- the fallback of the placement check, which no MI355X box takes by itself; 0
 * (default): as detected.  Same Sigma and L bit for bit. */
int covo_debug_set_ns_coherence(covo_handle_t h, int force_agent);
/* 0 makes the one-matrix Sigma chain evaluate its Rayleigh-Ritz pairs AFTER the squaring launch (ns_ritz_scan_kernel, what batches
 * and shared-device handles do) instead of inside it; default 1.  Same Sigma and L bit for bit: lambda_min is a function of the
 * matrix alone (sigma_ns.hip: ritz_decide).  2: a timing reference only -- rounds 1-4's rule (the Ritz step reads the filter's last
 * iterate and nothing else; Sigma then differs in the last bits). */
int covo_debug_set_ns_ritz_inside(covo_handle_t h, int on);
int covo_debug_hess_workspace(covo_handle_t h, double *out, int64_t offset_doubles, int64_t count, void *stream);
/* Test hook: `count` doubles at `offset_doubles` of the Hessians of the LAST covo_mpc_step_batched on this handle
 * ([n_envs][128][128], the Sigma chain's input), copied to the HOST buffer `out` (asynchronously on `stream`). */
int covo_debug_batched_hessians(covo_handle_t h, double *out, int64_t offset_doubles, int64_t count, void *stream);

/* Profiling aid: covo_sigma for ONE matrix that also stores shader-clock ticks (s_memtime) at the kernel's
 * phase boundaries into ticks_out (device uint64[32]): [0] start, [1] loaded+shifted, [2+i] end of sweep i,
 * [20] sweeps done, [21] spectrum map, [22] H H^T, [23] Cholesky, [24] number of sweeps. */
int covo_sigma_profile(covo_handle_t h, const double *R, float sample_sigma, float *Sigma_out, float *L_out,
                       uint64_t *ticks_out, void *stream);

/* Everything one controller __call__ does between "shift the mean" and "new mean", for this handle's shard
 * of samples, in ONE call (controllers/covo.py:201-275, mppi.py:43-125):
 *   shift mean -> [online: Hessian -> optimal Sigma -> Cholesky | offline: L_table[state.time] | mppi: shift
 *   a_cov, factor the 4x4 blocks] -> in-kernel Philox draw + noise GEMM -> fused rollout -> softmax reduction
 *   -> (partial_out == NULL) new mean written back to a_mean, or (partial_out != NULL) this shard's record.
 * All pointers are device pointers that must stay valid and UNCHANGED from call to call for the sequence to be
 * captured into a hipGraph (second identical call) and replayed (later calls); changing any of them or
 * `params` falls back to eager launches and re-captures.  key0/key1, f_disturb_shared and `state` may change freely. */
typedef struct covo_step_args {
    int32_t mode;            /* COVO_MODE_* */
    int32_t n_samples;       /* <= n_local */
    int32_t T;               /* rows of pos_traj / vel_traj */
    int32_t n_table;         /* offline: rows of L_table */
    const float *state;      /* float[32], the noisy state (controllers/covo.py:198).  Must stay valid and unchanged until the step
                              * has run: eager covo-online steps read it where it lies from EVERY launch (the begin work rides in the
                              * Hessian's first launch, covo_debug_set_fold_begin); other steps copy it in their begin launch */
    const float *pos_traj;
    const float *vel_traj;
    float *a_mean;           /* float[128] in/out */
    float *a_mean_shift;     /* nullable: float[128] receives the shifted old mean (needed by covo_merge's blend) */
    float *a_cov;            /* online: float[128][128] Sigma out (nullable); mppi: float[H][4][4] in/out (shifted) */
    const float *L_table;    /* offline: float[n_table][128][128] lower Cholesky factors of a_cov_offline */
    float *a;                /* work: float[H][n_samples][4] */
    float *cost;             /* work: float[n_samples] */
    float *groupmin;         /* work: float[ceil(n_samples/64)] */
    double *pos_stats;       /* nullable, as in covo_rollout_cost */
    float *partial_out;      /* nullable: this shard's record instead of finishing locally -- float[COVO_PARTIAL_FLOATS], or, MPPI with
                              *    gamma_sigma != 0, float[COVO_PARTIAL_FLOATS + COVO_COV_FLOATS] (with the second moments) */
    int64_t sample_offset;   /* global id of this shard's first sample */
    float gamma_mean;
    float sample_sigma;
    int32_t derive_keys;     /* 1: key0/key1 are the controller's raw rng_act; the sampling key (covo.py:212, mppi.py:53), the
                              *    rollouts' step key (covo.py:225, mppi.py:69) and everything drawn from it -- MPPI's shared
                              *    gaussian vector, the uniform draws of PERIODIC / MIXED, get_hessian's per-step keys
                              *    (covo.py:150-153, keyed by the raw rng_act) -- are derived on the device exactly as the Python
                              *    host does (random.py); f_disturb_shared is ignored.
                              * 0: key0/key1 = the sampling key; f_disturb_shared as given; PERIODIC / SIN / DRAG / MIXED need
                              *    derive_keys = 1 */
    int32_t rollout_deterministic; /* step_env's `deterministic` in the sampling rollouts: 1 for CoVO (covo.py:231), 0 for MPPI
                              *    (mppi.py:74): switches the GAUSSIAN model off (quadrotor.py:234-235) */
    float gamma_sigma;       /* MPPI: != 0 adapts a_cov in place after the mean update (mppi.py:119-125, covo_softmax_update_cov);
                              *    single shard */
    int32_t pad_;
    const float *a_mean_in;  /* nullable: the control_params.a_mean INPUT of this call when it does not live in `a_mean` (float[128],
                              *    read by the step's first launch only; covo.py:201 reads control_params.a_mean, :275 returns a new
                              *    one).  NULL: `a_mean` is read and then overwritten (a controller that carries its own mean).  Like
                              *    `state` it may change from call to call without invalidating the captured graph.  Must be NULL
                              *    for covo_run_episode (the episode carries the mean). */
} covo_step_args;

int covo_mpc_step(covo_handle_t h, const covo_env_params *params, const covo_step_args *args, uint32_t key0,
                  uint32_t key1, const float *f_disturb_shared /* [host float[3]] or NULL */, void *stream);

/* covo-online for n_envs INDEPENDENT env instances in one call / one hipGraph (BASELINE configs[4]; "replicas only": no
 * exchange between instances).  Every instance has its own noisy state, reference trajectory (T rows), parameters
 * (domain randomisation), mean and key; the Hessians and the Sigma chain of all instances run as ONE batched set of
 * launches, the sampling path (noise GEMM, rollout, softmax update) per instance.  Results per instance are bit-identical
 * to covo_mpc_step on that instance alone.  params: host array [n_envs]; keys: host uint32[n_envs][2], each the raw
 * rng_act of that instance's controller call (the sampling key is derived on the device, covo.py:212).  All instances
 * share reward_kind, rollover_terminate and disturb_kind (one kernel variant per launch); every disturbance model is
 * taken -- for PERIODIC / SIN / DRAG / MIXED the call builds each instance's per-step tables (covo_disturb_table's, SHARED
 * and HESSIAN key threading) from that instance's state, raw key and disturb_params inside the same graph. */
#define COVO_MAX_ENVS 64
typedef struct covo_batch_args {
    int32_t n_envs;          /* <= COVO_MAX_ENVS */
    int32_t n_samples;       /* per instance, <= n_local */
    int32_t T;               /* rows of every instance's pos_traj / vel_traj */
    int32_t pad_;
    const float *states;     /* float[n_envs][32] noisy states */
    const float *pos_traj;   /* float[n_envs][T][3] */
    const float *vel_traj;   /* float[n_envs][T][3] */
    float *a_mean;           /* float[n_envs][128] in/out */
    float *a_cov;            /* float[n_envs][128][128] Sigma out (nullable) */
    float *a;                /* work: float[n_envs][H][n_samples][4] */
    float *cost;             /* work: float[n_envs][n_samples] */
    float *groupmin;         /* work: float[n_envs][ceil(n_samples/64)] */
    float gamma_mean;
    float sample_sigma;
} covo_batch_args;

int covo_mpc_step_batched(covo_handle_t h, const covo_batch_args *args, const covo_env_params *params,
                          const uint32_t *keys, void *stream);

/* Lower Cholesky factors of `batch` symmetric PD n x n fp32 matrices (n <= 128), the
 * factorisation inside jax.random.multivariate_normal (covo.py:216, mppi.py:59). */
int covo_cholesky(covo_handle_t h, const float *A, int32_t n, int32_t batch, float *L_out, void *stream);

/* One closed-loop ENVIRONMENT step on the device: Quad3D.step_env + get_info for the controllers'
 * eval loop (envs/quadrotor.py:215-263, 314-361; dynamics/free.py:66-72, 114-202).  `state` (float[32], layout above) is
 * the true state, advanced in place; `noisy_state` receives the noisy copy of the NEW state (the controller's input);
 * `action` = float[4] on the device (the controller's u); `step_key` = uint32[2] on the HOST: the key Quad3D.step
 * receives -- the kernel derives the (disturbance, pos, vel, quat, omega) noise keys from it like the Python env; log (nullable) float[..][4] gets
 * {reward, err_pos, err_vel, done} of the PRE-step state at row log_index.  acc_traj: float[T][3].  Reward and disturbance
 * model (all six of free.py:9-72, the next step's force from the PRE-step state) follow params->reward_kind / disturb_kind.
 * AUTO-RESET (BaseEnvironment.step, quadjax/envs/base.py:22-40; params->reset_traj != COVO_TRAJ_NONE): when the PRE-step state is
 * terminal (quadrotor.py:479-490) the kernel stores what reset_env(key_reset) gives instead of the stepped state -- key_reset =
 * split(step_key)[1]; a NEW reference trajectory from params->reset_traj's generator written over pos_traj / vel_traj / acc_traj
 * (which are therefore written to, despite the const: T must be that generator's row count), the zero state with
 * f_disturb ~ U(-reset_disturb_scale, reset_disturb_scale), time 0, targets = row 0, and the noisy copy from reset_env's own
 * info key -- and the log row carries {reward of the pre-step state, err_pos, err_vel of the RESET state, 1}, which is what
 * eval_env records after env.step (quadrotor.py:531-538).  The controller's state carries on, as in the reference. */
int covo_env_step(covo_handle_t h, float *state, float *noisy_state, const float *pos_traj, const float *vel_traj,
                  const float *acc_traj, int32_t T, const covo_env_params *params, const float *action,
                  const uint32_t *step_key, int32_t noisy_on, float obs_noise_scale, float *log, int32_t log_index,
                  void *stream);

/* covo-offline's nominal trajectory (controllers/covo.py:58-99 with the PID law of controllers/pid.py:38-84 and the
 * expansion gains of covo.py:48-53): from `state0` (float[32], true reset state) `n_steps` PID-tracked,
 * NON-deterministic env steps (keys split from key0/key1 as the Python loop does) give states_out float[n_steps][32];
 * from each of them H deterministic PID steps give the nominal means a_means_out float[n_steps][128] -- the inputs of
 * the batched covo_hessian / covo_sigma that build the per-episode Sigma table.  `pid_params`: the parameters the PID
 * law uses (pid.py:33: the env's DEFAULT m, g, max_thrust, max_omega even under domain randomisation).  All disturbance
 * models of params->disturb_kind act in both loops (the nominal one is deterministic=True: GAUSSIAN off).
 * keys_out [DEVICE uint32[n_steps][2], nullable]: the scan's carry key at every start state = the key get_hessian receives
 * for that row (covo.py:77) -> covo_disturb_table(keys_dev = keys_out, COVO_DISTURB_KEYS_HESSIAN). */
int covo_pid_nominal(covo_handle_t h, const float *state0, const float *pos_traj, const float *vel_traj,
                     const float *acc_traj, int32_t T, const covo_env_params *params, const covo_env_params *pid_params,
                     float Kp, float Kd, float Kp_att, uint32_t key0, uint32_t key1, int32_t n_steps,
                     float *states_out, float *a_means_out, uint32_t *keys_out, void *stream);

/* A whole closed-loop episode segment with no host work between the steps (the reference traces
 * eval_env's run_one_step into one XLA program, envs/quadrotor.py:506-591): n_steps x { covo_mpc_step on the noisy state
 * -> covo_env_step with u = a_mean[0] }, keys threaded as run_one_step does (rng, rng_act, rng_step, _ = split(rng, 4);
 * rng, _ = split(rng)).  `args` as for covo_mpc_step with derive_keys = 1, partial_out = NULL and args->state = the
 * NOISY state buffer (float[32], rewritten by every env step); state_true float[32]; rng uint32[2] in/out (host);
 * log float[n_steps][4] (nullable).  Asynchronous: returns after enqueueing; one sync at the end of the episode. */
int covo_run_episode(covo_handle_t h, const covo_env_params *params, const covo_step_args *args, float *state_true,
                     const float *acc_traj, int32_t noisy_on, float obs_noise_scale, float *log, uint32_t *rng,
                     int32_t n_steps, void *stream);

/* BASELINE configs[4] end to end: the env step and the episode driver for n_envs INDEPENDENT, domain-randomised env instances
 * (quadrotor.py:132-171 samples each instance's parameters; :506-591 is the per-instance eval loop, which the reference reaches
 * through jax.vmap).  covo_env_step_batched = covo_env_step for every instance in ONE launch (workgroup e = instance e: its own
 * true state states[e], noisy copy noisy[e], trajectories [e][T][3], action a_mean[e][0..3], parameters params[e], key
 * step_keys[e]; log float[n_envs][log_stride][4], nullable); instance e's step is bit-identical to covo_env_step on it alone.
 * All instances share reward_kind, rollover_terminate, max_steps_in_episode and disturb_kind (as in covo_mpc_step_batched).
 * covo_run_episode_batched = n_steps x { covo_mpc_step_batched on the noisy states -> covo_env_step_batched }, every instance's
 * key chain threaded like run_one_step (rngs: host uint32[n_envs][2], in/out); args->states = the noisy states (rewritten by
 * every env step).  Asynchronous; one host sync per episode segment.  "Replicas only": instances shard over ranks with no
 * collective (bench.py --config envs). */
int covo_env_step_batched(covo_handle_t h, int32_t n_envs, float *states, float *noisy_states, const float *pos_traj,
                          const float *vel_traj, const float *acc_traj, int32_t T, const covo_env_params *params /* [n_envs] */,
                          const float *a_mean /* [n_envs][128] */, const uint32_t *step_keys /* host [n_envs][2] */,
                          int32_t noisy_on, float obs_noise_scale, float *log, int32_t log_stride, int32_t log_index, void *stream);
int covo_run_episode_batched(covo_handle_t h, const covo_batch_args *args, const covo_env_params *params /* [n_envs] */,
                             float *states_true /* [n_envs][32] */, const float *acc_traj /* [n_envs][T][3] */, int32_t noisy_on,
                             float obs_noise_scale, float *log /* [n_envs][log_stride][4], nullable */, int32_t log_stride,
                             int32_t log_index, uint32_t *rngs /* host [n_envs][2], in/out */, int32_t n_steps, void *stream);

/* Profiling aid: `reps` copies of the selected launches of one control step, captured into one hipGraph and
 * replayed; *us_out = GPU microseconds per copy.  step_mask bits: 1 shift_mean, 2 Hessian, 4 Sigma, 8 noise GEMM,
 * 16 rollout, 32 softmax update; hess_mask bits: the four kernels of the adjoint Hessian; sigma_stages 1..4:
 * prep+squarings, +Ritz, +Newton-Schulz, +finalize.  Uses the buffers in `args` exactly like covo_mpc_step -- and, like a
 * real step, MUTATES the controller state they hold: every replayed copy that includes the update writes args->a_mean, MPPI's
 * begin work / fused small step shifts args->a_cov in place once per copy (plus once for the scratch refresh an eager handle
 * needs first), and the handle's sequence number advances.  Time on a state you can afford to lose, or snapshot a_mean / a_cov
 * around the call (bench.py times after its value-bearing steps). */
int covo_debug_time_step(covo_handle_t h, const covo_env_params *params, const covo_step_args *args, int32_t step_mask,
                         int32_t hess_mask, int32_t sigma_stages, int32_t reps, float *us_out, void *stream);

/* The same for the LAST covo_mpc_step_batched call of the handle (all instances; the begin launch, bit 1, re-splits the keys and
 * is normally left out). */
int covo_debug_time_batched(covo_handle_t h, int32_t step_mask, int32_t reps, float *us_out, void *stream);

/* Test hook: a one-thread kernel on `stream` ORs `bits` into the handle's status word the way a failing kernel would
 * (device store to host-mapped memory); covo_device_status / COVO_E_DEVICE can then be exercised without starving a barrier. */
int covo_debug_raise_device_status(covo_handle_t h, int32_t bits, void *stream);

/* Profiling aid: covo_rollout_cost `reps` times back to back on `stream` between two events, three batches; us_out[0] =
 * mean GPU microseconds per launch over all batches, us_out[1] = the fastest batch's (what bench.py reports as
 * roofline.launch_us / launch_us_min: a Python loop of single calls is host-bound below ~10 us per launch).
 * with_records != 0: the variant covo_mpc_step runs -- every workgroup also leaves its online-softmax record
 * (rollout_pipe3_kernel<..., REC = true>).  Synchronises the stream.  Other arguments as covo_rollout_cost (no position
 * statistics). */
int covo_debug_time_rollout(covo_handle_t h, const float *state, const float *pos_traj, const float *vel_traj, int32_t T,
                            const covo_env_params *params, const float *f_disturb_shared, const float *f_disturb_steps,
                            const float *a, int32_t N,
                            float *cost_out, float *groupmin, int32_t with_records, int32_t reps, float *us_out /* [host float[2]] */,
                            void *stream);

#ifdef __cplusplus
}
#endif
#endif /* COVO_HIP_H */
