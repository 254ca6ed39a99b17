// covo_common.hpp -- internal declarations shared by the gfx950 kernels and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/covo_hip.h"
#include "quad_model.hpp"

#define COVO_WAVE 64

// state buffer offsets (include/covo_hip.h "Data layouts")
enum { ST_POS = 0, ST_VEL = 3, ST_QUAT = 6, ST_OMEGA = 10, ST_FDIST = 13, ST_POSTAR = 16, ST_VELTAR = 19, ST_ACCTAR = 22, ST_TIME = 25 };

// Experiment switches of ONE handle (round 6: rounds 1-5 kept them in process-global variables that flipped every handle of the
// process and were not thread-safe).  covo_create fills them from the environment (covo_default_opts: COVO_STREAM_GEMM,
// COVO_FUSE_SMALL, COVO_FOLD_BEGIN, COVO_NS_DEFLATE, COVO_NS_RITZ_INSIDE, read once per handle); covo_debug_set_*(handle, ...) change
// them for that handle only and bump `epoch`, which makes the handle re-capture its step graphs (they bake the launch set in).
struct CovoOpts {
    int stream_gemm;   // covo-online's noise GEMM streamed inside the Sigma chain's finalize launch (sigma_ns.hip: ns_finalize_stream_kernel)
    int fuse_small;    // the fused small step (step_small.hip) is taken where eligible
    int fold_begin;    // eager covo-online steps: the begin work rides in the Hessian's first launch
    int ns_tail_iters, ns_tail_squarings, ns_tail_iters_batched, ns_tail_squarings_batched;  // phases folded into the persistent launches
    int ns_deflate;      // the Newton-Schulz iteration deflates the bottom eigenpair
    int ns_force_agent;  // take the agent-scope coherence fallback although the placement check passed
    int ns_merged;       // one matrix: squaring chain + evaluations + Newton-Schulz iterations as ONE launch (ns_chain_kernel)
    int ns_ritz_inside;  // 1: the Rayleigh-Ritz evaluations ride in the squaring launch; 0: one scan launch; 2: the last iterate only
    int epoch;
};
CovoOpts covo_default_opts();  // step.hip

struct covo_ctx {
    covo_config cfg;
    CovoOpts opt;
    int device;
    // workspace (device)
    float *ws_partials;   // [max_blocks][COVO_PARTIAL_FLOATS] stage-1 records of the softmax reduce
    float *ws_partials_cov;  // [max_blocks][452] stage-1 records with second moments (MPPI covariance adaptation, reduce.hip)
    float *ws_blockmin;   // [ceil(n_local/64)] per-wave cost minima when the caller passes none
    double *ws_stats;     // [ceil(n_local/256)][H*6] per-block position statistics
    void *ws_sigma;       // scratch of the eigh-free Sigma pipeline (grown on demand, outside graph capture)
    size_t ws_sigma_bytes;
    void *ws_hess;        // scratch of the second-order-adjoint Hessian (grown on demand, outside graph capture)
    size_t ws_hess_bytes;
    void *step;               // StepState (step.hip): fused-step scratch + graph cache
    void *batch;              // BatchState (step.hip): env-batched step scratch + graph cache
    hipStream_t side_stream;  // forked work inside one call (joined before the call's last kernel)
    hipEvent_t ev_fork, ev_join;
    int max_red_blocks;
    int *status_host;         // host-mapped sticky status word (COVO_DEVSTAT_* bits written by kernels), see covo_device_status
    int *status_dev;          // its device address
    void *exchange;           // Exchange (exchange.hip): peer-write exchange of the rank records, or null
    int dbg_epoch;            // opt.epoch when this handle's step graphs were captured (a debug setter since then: re-capture)
};

void covo_set_error(const char *fmt, ...);

// hipFuncSetAttribute is per DEVICE: a one-time opt-in (dynamic LDS above 64 KB) guarded by a process-wide `static bool` was set on
// the first device only.  Returns true the first time it is called for the current device with this mask.
static inline bool covo_first_on_device(unsigned long long &mask)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (mask & bit) return false;
    mask |= bit;
    return true;
}

// A pointer a kernel LOADS from memory (a field of a per-instance argument block, a table of buffers) is a GENERIC pointer to the
// compiler: every access through it becomes a flat_load / flat_store, which counts on BOTH vmcnt and lgkmcnt -- an
// `s_waitcnt lgkmcnt(0)` in front of an LDS hand-off or a barrier then also drains the loads in flight (round 6: the env-batched
// rollout kept its 12 prefetched stripes for two steps instead of twelve and ran 30 % slower than the same kernel taking its
// pointers as kernel arguments, which clang knows to be global).  rebase_global re-expresses such a pointer as
// (a kernel-argument pointer of the same kind) + (byte distance between the two): the address space then follows from the kernel
// argument.  The distance goes through an empty asm, else InstCombine folds base + (p - base) back into p.  Neither a cast
// through address space 1 and back nor an llvm.assume(!is.shared && !is.private) survives to InferAddressSpaces here (ROCm 7.2).
template <class T>
__device__ __forceinline__ T *rebase_global(T *kernarg_base, T *loaded)
{
    long long off = (long long)(uintptr_t)loaded - (long long)(uintptr_t)kernarg_base;
    int lo = (int)off, hi = (int)(off >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    off = ((long long)hi << 32) | (unsigned)lo;
    return (T *)((const char *)kernarg_base + off);
}

// debug/profiling switches (covo_debug_set): which launches of the Hessian (bit k = kernel k of hessian_adj.hip) and how
// many stages of the Sigma pipeline (1 prep+squarings, 2 +Ritz, 3 +Newton-Schulz, 4 +finalize) are enqueued.
// Defaults enqueue everything; only covo_debug_time_step changes them, and restores them.
extern int g_dbg_hess_mask, g_dbg_sigma_stages;
void sigma_ns_tail_defaults(CovoOpts &o);  // sigma_ns.hip: the four tail lengths at their defaults (defined in ONE place)

#define COVO_CHECK_HIP(expr)                                                         \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            covo_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return (int)_e;                                                          \
        }                                                                            \
    } while (0)

template <class U>
__host__ __device__ inline qm::Consts<U> make_consts(const covo_env_params &p)
{
    qm::Consts<U> c;
    // the model constants are the fp32 parameter values (JAX default dtype); derived constants are
    // formed in the working precision U from them (U = float reproduces the reference's fp32 products)
    c.thrust_half = U(0.5) * U(p.max_thrust) * U(p.action_scale);
    for (int i = 0; i < 3; ++i) c.komega[i] = U(p.max_omega[i]) * U(p.action_scale);
    c.dt = U(p.dt);
    c.half_dt = U(0.5) * U(p.dt);
    c.neg_g = -U(p.g);
    c.inv_m = U(1) / U(p.m);
    c.alpha = U(p.alpha_bodyrate);
    c.one_m_alpha = U(1) - U(p.alpha_bodyrate);
    c.pos_limit = U(p.pos_limit);
    return c;
}

// ---- wave-level reductions over 64 lanes (DPP-lowered by the compiler from __shfl_xor)
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// launch entry points implemented in the .hip files (host functions)
int launch_randn(uint32_t k0, uint32_t k1, int64_t off, int n_samples, int n_cols, float *out, hipStream_t s);
int launch_randn_jax(uint32_t k0, uint32_t k1, int64_t n_total, int64_t off, int n_samples, int mppi, float *out, hipStream_t s);
// eps == null: epsilon is drawn in-kernel from (k0, k1, sample_offset + n) (rng_device.hpp)
// cov != null (fused covo-online step): a_cov = cz sym(Z) is NOT written by the chain's finalize launch -- one workgroup that
// everything after it waits for -- but by the first workgroups of the noise GEMM that follows (same expression, same bits);
// launch_sigma_ns fills *cov with where Z, its transpose and the scalars live.
// threads per workgroup the noise GEMM launches with for (N, batch) (noise_gemm.hip: 512 once a launch fills the chip, else 256);
// the rollout's XCD-affine sample mapping follows it (rollout.hip: xcd_remap = 64-sample groups per GEMM workgroup)
int noise_gemm_block_threads(int N, int batch);
int noise_gemm_groups_per_workgroup(int N, int batch);
struct CovDeferred {
    const double *Z[2], *Zt[2];  // the two Newton-Schulz buffers of Z and of its stored transpose
    const double *zbuf;          // != 0: buffer 1 holds the final iterate
    const double *cz;            // Sigma = cz sym(Z) (NaN when a grid barrier of the chain timed out)
    const double *zcoef, *u;     // deflated chain (sigma_ns.hip): Z = Z~ + zcoef u u^T (*zcoef == 0: off)
    float *out;                  // a_cov [128][128]; null: nothing deferred
};
int launch_noise_gemm(const float *L, const float *mu, const float *eps, uint32_t k0, uint32_t k1, int64_t sample_offset,
                      int N, float *a, hipStream_t s, const uint32_t *dyn = nullptr, const float *state_for_time = nullptr,
                      int n_table = 0, int batch = 1,  // batch > 1 (in-kernel Philox only): dense per-instance L, mu, dyn, a
                      bool eps_tiled = false, const CovDeferred *cov = nullptr,        // eps is the tile-ordered image of eps_tiles.hpp
                      bool propagate_nan = false);                                     // COVO_FLAG_PROPAGATE_NAN: jnp.clip's NaN semantics
int launch_noise_blockdiag(const float *Ls, const float *mu, const float *eps, uint32_t k0, uint32_t k1,
                           int64_t sample_offset, int N, float *a, hipStream_t s, const uint32_t *dyn = nullptr,
                           bool propagate_nan = false);
static inline bool covo_propagate_nan(const covo_ctx *h) { return (h->cfg.flags & COVO_FLAG_PROPAGATE_NAN) != 0; }
int launch_rollout(const float *state, const float *pos_traj, const float *vel_traj, int T, const covo_env_params &p,
                   const float *f_shared, const float *a, int N, float discount, bool trust_clipped, float *cost,
                   float *groupmin, double *pos_stats, double *stats_ws, hipStream_t s, const float *f_shared_dev = nullptr,
                   float *records = nullptr, float lam = 0.0f,   // records: one online-softmax record per workgroup (rollout.hip)
                   const float *f_tab = nullptr,                 // [H][4] per-step disturbance table (disturb.hip), device
                   int xcd_groups = 0,    // 64-sample groups per workgroup of the kernel that wrote `a` (0: the noise GEMM's for this N)
                   bool propagate_nan = false);  // the re-clip of untrusted stripes keeps a NaN (COVO_FLAG_PROPAGATE_NAN)
int launch_disturb_table(const covo_env_params &p, const float *state, int batch, const uint32_t *keys_dev, uint32_t key0,
                         uint32_t key1, int key_mode, int deterministic, float *out, hipStream_t s);
int launch_disturb_tables_step(const covo_env_params &p, const float *state, const uint32_t *dyn, int rollout_deterministic,
                               float *tab_rollout, float *tab_hess, hipStream_t s);
// env-batched step: per-instance models (dm::Model[n], host-filled, copied to the device by the caller) and both tables of
// every instance in one launch ([n][H][4] each; raw keys at dyn[12 e + 10..11])
size_t disturb_models_bytes(int n);
void disturb_fill_models(const covo_env_params *params, int n, void *out);
int launch_disturb_tables_batched(const void *models_dev, const float *states, const uint32_t *dyn, int n_envs,
                                  int rollout_deterministic, float *tab_rollout, float *tab_hess, hipStream_t s);
int rollout_workgroups(int N, bool stats, int nbatch = 1);
size_t rollout_args_bytes(int n);
void rollout_fill_args(void *out, int index, const float *state, const float *pos_traj, const float *vel_traj, int T,
                       const covo_env_params &p, const float *a, int N, float discount, float *cost, float *groupmin,
                       const float *f_shared_dev, float *records = nullptr, float lam = 0.0f, bool trust_clipped = true,
                       const float *f_tab = nullptr);
int launch_rollout_batched(const void *args_host, const void *args_dev, int nbatch, hipStream_t s);
// a_mean_out != null: finish on this GPU (normalise + blend); else write the merged record to partial_out
int launch_softmax_reduce(covo_ctx *h, const float *cost, const float *a, int N, const float *blockmin, int n_blockmin,
                          float *partial_out, const float *a_mean_old, float gamma_mean, float *a_mean_out,
                          hipStream_t s, float *partials_ws = nullptr, int batch = 1);  // batch > 1: dense per-instance slices, own partials_ws
// a_mean_out == null: the merged record goes to partial_out (sample-sharded step); batch > 1: dense per-instance slices
int launch_merge(const float *partials, int G, float lam, const float *a_mean_old, float gamma_mean, float *a_mean_out,
                 hipStream_t s, float *partial_out = nullptr, int batch = 1, int stride = COVO_PARTIAL_FLOATS);
// exchange.hip: the rank records of a sample-sharded step and their peer-write exchange
int launch_rank_stats_sum(const float *records, int G, double *out, hipStream_t s, bool cov = false);
int exchange_create(covo_ctx *h, int world, int rank, void *handle_out);
int exchange_connect(covo_ctx *h, const void *handles);
int exchange_set_timeout(covo_ctx *h, double seconds);
void exchange_destroy(covo_ctx *h);
bool exchange_ready(const covo_ctx *h);
int exchange_world(const covo_ctx *h);
int exchange_records(covo_ctx *h, const float *record, float *gathered_dst, const float **gathered_out, hipStream_t s,
                     int nfloats = COVO_RANK_RECORD_FLOATS);
size_t softmax_cov_workspace_floats(int max_blocks);
int launch_softmax_update_cov(covo_ctx *h, const float *cost, const float *a, int N, const float *blockmin, int n_blockmin,
                              const float *a_mean_old, float gamma_mean, const float *a_cov_old, float gamma_sigma,
                              float *a_mean_out, float *a_cov_out, hipStream_t s);
int launch_softmax_reduce_cov(covo_ctx *h, const float *cost, const float *a, int N, const float *blockmin, int n_blockmin,
                              const float *a_mean_old, float *record_out, hipStream_t s);
int launch_merge_cov(const float *records, int G, int stride, float lam, const float *a_mean_old, float gamma_mean,
                     const float *a_cov_old, float gamma_sigma, float *a_mean_out, float *a_cov_out, hipStream_t s);
int launch_shift_mean(const float *in, float *out, hipStream_t s);
size_t hessian_workspace_bytes(int batch);
struct SymStatsOut;  // sym_stats.hpp
// the step's begin work folded into the Hessian's first launch (eager covo-online steps; hessian_adj.hip: AdjArgs): the
// caller's unshifted mean, where the per-step scalars and the sequence number live, and the 48-byte block of step_begin.hpp
struct HessBegin {
    const float *a_mean_raw;
    uint32_t *dyn_out;
    unsigned *seq;
    const void *blk;  // DynBlock
    int derive_keys;
    float shared_noise_scale;
};
int launch_hessian(const float *state, const float *pos_traj, const float *vel_traj, int T, const covo_env_params &p,
                   const float *a_mean, int batch, double *R, void *workspace, hipStream_t s, const void *consts_dev = nullptr,
                   size_t traj_stride = 0,
                   const SymStatsOut *stats = nullptr,   // KD also leaves the Sigma chain's input statistics (sym_stats.hpp)
                   const float *f_tab = nullptr,         // [batch][H][4] per-step disturbance table (disturb.hip), device
                   const void *models_dev = nullptr,     // dm::Model[batch] next to consts_dev (drag / mixed with per-instance parameters)
                   int *status_dev = nullptr,            // the handle's sticky status word: COVO_DEVSTAT_ADJOINT on a costate time-out
                   const HessBegin *begin = nullptr);    // batch 1: KB also does the step's begin work (a_mean = where the shifted mean goes)
// true: launch_hessian leaves R's Sigma-chain statistics when asked to (the adjoint kernels do, for every disturbance model;
// launch_hessian_pairs does not)
inline bool hessian_leaves_stats(const covo_env_params &p)
{
    (void)p;
    return true;
}
size_t hessian_consts_bytes(int n);
void hessian_fill_consts(const covo_env_params *params, int n, void *out);
// the per-pair hyper-dual rollout version (hessian.hip): slower, independent derivation, kept as a cross-check
int launch_hessian_pairs(const float *state, const float *pos_traj, const float *vel_traj, int T, const covo_env_params &p,
                         const float *a_mean, int batch, double *R, hipStream_t s, const float *f_tab = nullptr,
                         const void *consts_dev = nullptr, size_t traj_stride = 0, const void *models_dev = nullptr);
int launch_sigma(const double *R, int batch, float sample_sigma, float *Sigma, float *L, unsigned long long *prof,
                 hipStream_t s);
size_t sigma_ns_workspace_bytes(int batch);
struct EpsGenArgs;  // eps_tiles.hpp
// gen != null (fused step): the finalize launch also draws the step's epsilon in tile order (eps_tiles.hpp)
// status: the handle's sticky status word (a timed-out grid barrier raises COVO_DEVSTAT_GRID_BARRIER there, next to the NaN
// outputs); persistent_ok = false (COVO_FLAG_SHARED_DEVICE): every phase its own launch
// r_has_stats (fused steps): R is exactly symmetric and its statistics (sym_stats.hpp) are already in the workspace -- left
// there by the Hessian's last launch through sigma_ns_stats_out -- so the chain starts without its prep launch
// stream != null (fused covo-online step, one matrix, persistent launches allowed): the noise GEMM of the step is carried out INSIDE
// the finalize launch, streamed under the factorisation (sigma_ns.hip: ns_finalize_stream_kernel); *streamed says whether it was
// (else the caller launches the GEMM).  gen / cov / Sigma / L are then unused: a_cov goes to stream->a_cov_out, the factor to
// stream->L_stream.
struct StreamGemmArgs {
    const float *mu;        // the shifted mean [128]
    const uint32_t *dyn;    // {key0, key1}: the step's sampling key in device memory
    int64_t sample_offset;  // global id of local sample 0
    int N;                  // samples of this shard
    float *a_out;           // [H][N][4] action stripes
    float *L_stream;        // [128][128] fp32: chol(Sigma), written panel by panel (write-through) while the workers read it
    unsigned *sync;         // [16]: [0] the step's sequence number (bumped by the begin launch), [1 + p]: panel p (16 columns) of
                            // L_stream is complete for the sequence number it holds
    float *a_cov_out;       // nullable: a_cov [128][128]
    int nanp;               // COVO_FLAG_PROPAGATE_NAN
};
int launch_sigma_ns(const CovoOpts &opt, const double *R, int batch, float sample_sigma, float *Sigma, float *L, void *workspace,
                    hipStream_t s, const EpsGenArgs *gen = nullptr, int *status = nullptr, bool persistent_ok = true,
                    CovDeferred *cov = nullptr, bool r_has_stats = false, const StreamGemmArgs *stream = nullptr,
                    bool *streamed = nullptr);
struct SymStatsOut;  // sym_stats.hpp
SymStatsOut sigma_ns_stats_out(void *workspace, int batch = 1);
void step_state_destroy(covo_ctx *h);
void step_graphs_drop(covo_ctx *h);  // before re-allocating h->ws_sigma / h->ws_hess: captured graphs hold their addresses
void batch_state_destroy(covo_ctx *h);
int covo_debug_time_batched_impl(covo_ctx *h, int step_mask, int reps, float *us_out, hipStream_t run);
int covo_step_batched_impl(covo_ctx *h, const covo_batch_args *args, const covo_env_params *params, const uint32_t *keys,
                           hipStream_t s);
int covo_step_impl(covo_ctx *h, const covo_env_params *params, const covo_step_args *args, uint32_t key0, uint32_t key1,
                   const float *f_shared, hipStream_t s);
int launch_cholesky(const float *A, int n, int batch, float *L, hipStream_t s);
size_t env_step_inst_bytes(int n);
void env_step_fill_inst(const covo_env_params *params, int n, void *out);  // host: EnvInst[n] (to be copied to the device)
int launch_env_step_batched(float *states, float *noisy, const float *pos_traj, const float *vel_traj, const float *acc_traj, int T,
                            const covo_env_params &params0, const void *inst_dev, int E, const float *a_mean,
                            const uint32_t *step_keys, int noisy_on, float obs_noise_scale, float *log, int log_stride,
                            int log_index, hipStream_t s);
int batch_env_inst(covo_ctx *h, const covo_env_params *params, int E, hipStream_t s, const void **inst_dev);  // step.hip
int launch_env_step(float *state, float *noisy, const float *pos_traj, const float *vel_traj, const float *acc_traj, int T,
                    const covo_env_params &p, const float *action, const uint32_t *step_key, int noisy_on,
                    float obs_noise_scale, float *log, int log_index, hipStream_t s);
int launch_pid_nominal(const float *state0, const float *pos_traj, const float *vel_traj, const float *acc_traj, int T,
                       const covo_env_params &p, const covo_env_params &pid_params, float Kp, float Kd, float Kp_att,
                       uint32_t key0, uint32_t key1, int n_steps, float *states, float *a_means, uint32_t *keys_out,
                       hipStream_t s);
int covo_debug_time_step_impl(covo_ctx *h, const covo_env_params *params, const covo_step_args *args, int step_mask,
                              int hess_mask, int sigma_stages, int reps, float *us_out, hipStream_t run);
