// capi.hip -- the extern "C" surface declared in include/covo_hip.h.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "covo_common.hpp"

static thread_local char g_err[512] = "";

void covo_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define REQUIRE(cond, ...)            \
    do {                              \
        if (!(cond)) {                \
            covo_set_error(__VA_ARGS__); \
            return COVO_E_BADARG;     \
        }                             \
    } while (0)

// a kernel of an earlier call raised a sticky status bit (covo_device_status): refuse to pile more work on poisoned data
#define CHECK_DEVICE(h, what)                                                                                                \
    do {                                                                                                                     \
        const int st_ = *(volatile int *)(h)->status_host;                                                                   \
        if (st_ != 0) {                                                                                                      \
            covo_set_error("%s: device status 0x%x from an earlier call%s%s (covo_device_status)", what, st_,               \
                           (st_ & COVO_DEVSTAT_EXCHANGE) ? ": a peer's rank record did not arrive (covo_exchange_records)"    \
                           : (st_ & COVO_DEVSTAT_ADJOINT) ? ": the adjoint Hessian's costate wait timed out (that call's R / Sigma / L are NaN)" : "", \
                           (st_ & COVO_DEVSTAT_GRID_BARRIER)                                                                 \
                               ? ": a grid barrier of the Sigma chain timed out -- the GPU is shared with other work; that " \
                                 "call's Sigma / L / mean are NaN.  Create the handle with COVO_FLAG_SHARED_DEVICE"        \
                               : "");                                                                                        \
            return COVO_E_DEVICE;                                                                                            \
        }                                                                                                                    \
    } while (0)

// reward / disturbance selectors of covo_env_params
static int check_model(const covo_env_params *p, const char *what)
{
    if (p->reward_kind != COVO_REWARD_PENYAW && p->reward_kind != COVO_REWARD_REALWORLD) {
        covo_set_error("%s: reward_kind=%d (COVO_REWARD_*)", what, p->reward_kind);
        return COVO_E_BADARG;
    }
    if (p->disturb_kind < COVO_DISTURB_NONE || p->disturb_kind > COVO_DISTURB_MIXED) {
        covo_set_error("%s: disturb_kind=%d (COVO_DISTURB_*)", what, p->disturb_kind);
        return COVO_E_BADARG;
    }
    if (p->disturb_kind >= COVO_DISTURB_PERIODIC && p->disturb_period <= 0) {
        covo_set_error("%s: disturb_period=%d", what, p->disturb_period);
        return COVO_E_BADARG;
    }
    return 0;
}
static bool needs_table(const covo_env_params *p) { return p->disturb_kind >= COVO_DISTURB_PERIODIC && p->disturb_kind <= COVO_DISTURB_MIXED; }
#define CHECK_MODEL(p, what)               \
    do {                                   \
        const int rc_ = check_model(p, what); \
        if (rc_) return rc_;               \
    } while (0)

__global__ void raise_status_kernel(int *status, int bits)
{
    __hip_atomic_fetch_or(status, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

int covo_debug_batched_hessians_impl(covo_ctx *h, double *out, int64_t offset_doubles, int64_t count, hipStream_t s);  // step.hip

extern "C" {

const char *covo_last_error(void) { return g_err; }
int covo_abi_version(void) { return COVO_ABI_VERSION; }

int covo_create(const covo_config *cfg, covo_handle_t *out)
{
    REQUIRE(cfg && out, "covo_create: null argument");
    REQUIRE(cfg->H == COVO_H && cfg->du == COVO_DU, "covo_create: H=%d du=%d unsupported (kernels are built for H=%d du=%d)",
            cfg->H, cfg->du, COVO_H, COVO_DU);
    REQUIRE(cfg->n_local > 0, "covo_create: n_local=%d", cfg->n_local);
    REQUIRE(cfg->lam > 0.0f, "covo_create: lam=%g", (double)cfg->lam);
    covo_ctx *h = new covo_ctx();
    h->opt = covo_default_opts();
    h->dbg_epoch = h->opt.epoch;
    h->cfg = *cfg;
    COVO_CHECK_HIP(hipGetDevice(&h->device));
    h->max_red_blocks = 256;
    h->exchange = nullptr;
    const int nb = (cfg->n_local + 255) / 256, ng = (cfg->n_local + 63) / 64;
    COVO_CHECK_HIP(hipMalloc(&h->ws_partials, (size_t)h->max_red_blocks * COVO_PARTIAL_FLOATS * sizeof(float)));
    COVO_CHECK_HIP(hipMalloc(&h->ws_partials_cov, softmax_cov_workspace_floats(h->max_red_blocks) * sizeof(float)));
    COVO_CHECK_HIP(hipMalloc(&h->ws_blockmin, (size_t)ng * sizeof(float)));
    COVO_CHECK_HIP(hipMalloc(&h->ws_stats, (size_t)(nb > 256 ? nb : 256) * COVO_H * 6 * sizeof(double)));  // one row per rollout workgroup
    h->ws_sigma_bytes = sigma_ns_workspace_bytes(1);
    COVO_CHECK_HIP(hipMalloc(&h->ws_sigma, h->ws_sigma_bytes));
    h->ws_hess_bytes = hessian_workspace_bytes(1);
    COVO_CHECK_HIP(hipMalloc(&h->ws_hess, h->ws_hess_bytes));
    COVO_CHECK_HIP(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
    COVO_CHECK_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    COVO_CHECK_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    COVO_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->status_host), sizeof(int), hipHostMallocMapped));
    *h->status_host = 0;
    COVO_CHECK_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->status_dev), h->status_host, 0));
    *out = h;
    return 0;
}

int covo_destroy(covo_handle_t h)
{
    if (!h) return COVO_E_NOHANDLE;
    step_state_destroy(h);
    batch_state_destroy(h);
    exchange_destroy(h);
    int rc = 0;
#define DESTROY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess && !rc) {                                                                  \
            covo_set_error("covo_destroy: %s failed: %s", #expr, hipGetErrorString(_e));               \
            rc = (int)_e;                                                                               \
        }                                                                                               \
    } while (0)
    DESTROY(hipFree(h->ws_partials));
    DESTROY(hipFree(h->ws_partials_cov));
    DESTROY(hipFree(h->ws_blockmin));
    DESTROY(hipFree(h->ws_stats));
    DESTROY(hipFree(h->ws_sigma));
    DESTROY(hipFree(h->ws_hess));
    DESTROY(hipEventDestroy(h->ev_fork));
    DESTROY(hipEventDestroy(h->ev_join));
    DESTROY(hipStreamDestroy(h->side_stream));
    DESTROY(hipHostFree(h->status_host));
#undef DESTROY
    (void)hipGetLastError();  // never leave a sticky error behind for the caller's runtime (torch checks it)
    delete h;
    return rc;
}

int covo_device_status(covo_handle_t h, int32_t clear)
{
    if (!h) return COVO_E_NOHANDLE;
    const int st = *(volatile int *)h->status_host;
    if (clear) *(volatile int *)h->status_host = 0;
    return st;
}

int covo_debug_raise_device_status(covo_handle_t h, int32_t bits, void *stream)
{
    REQUIRE(h, "covo_debug_raise_device_status: null handle");
    hipLaunchKernelGGL(raise_status_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, h->status_dev, (int)bits);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}

int covo_randn(covo_handle_t h, uint32_t key0, uint32_t key1, int64_t sample_offset, int32_t n_samples, int32_t n_cols,
               float *eps_out, void *stream)
{
    REQUIRE(h, "covo_randn: null handle");
    REQUIRE(eps_out && n_samples > 0 && n_cols > 0, "covo_randn: bad argument");
    return launch_randn(key0, key1, sample_offset, n_samples, n_cols, eps_out, (hipStream_t)stream);
}

int covo_randn_jax(covo_handle_t h, uint32_t key0, uint32_t key1, int64_t n_total, int64_t sample_offset, int32_t n_samples,
                   int32_t mppi, float *eps_out, void *stream)
{
    REQUIRE(h, "covo_randn_jax: null handle");
    REQUIRE(eps_out && n_samples > 0 && sample_offset >= 0 && n_total >= sample_offset + n_samples, "covo_randn_jax: bad argument");
    REQUIRE(n_total < (1ll << 31), "covo_randn_jax: n_total=%lld: jax's iota(2 N) counters are 32-bit", (long long)n_total);
    return launch_randn_jax(key0, key1, n_total, sample_offset, n_samples, mppi, eps_out, (hipStream_t)stream);
}

int covo_noise_gemm(covo_handle_t h, const float *L, const float *mu, const float *eps, int32_t N, float *a_out,
                    void *stream)
{
    REQUIRE(h, "covo_noise_gemm: null handle");
    CHECK_DEVICE(h, "covo_noise_gemm");
    REQUIRE(L && mu && eps && a_out && N > 0, "covo_noise_gemm: bad argument");
    return launch_noise_gemm(L, mu, eps, 0u, 0u, 0, N, a_out, (hipStream_t)stream, nullptr, nullptr, 0, 1, false, nullptr,
                             covo_propagate_nan(h));
}

int covo_noise_gemm_philox(covo_handle_t h, const float *L, const float *mu, uint32_t key0, uint32_t key1,
                           int64_t sample_offset, int32_t N, float *a_out, void *stream)
{
    REQUIRE(h, "covo_noise_gemm_philox: null handle");
    CHECK_DEVICE(h, "covo_noise_gemm_philox");
    REQUIRE(L && mu && a_out && N > 0, "covo_noise_gemm_philox: bad argument");
    return launch_noise_gemm(L, mu, nullptr, key0, key1, sample_offset, N, a_out, (hipStream_t)stream, nullptr, nullptr, 0, 1, false,
                             nullptr, covo_propagate_nan(h));
}

int covo_noise_blockdiag(covo_handle_t h, const float *Ls, const float *mu, const float *eps, int32_t N, float *a_out,
                         void *stream)
{
    REQUIRE(h, "covo_noise_blockdiag: null handle");
    REQUIRE(Ls && mu && eps && a_out && N > 0, "covo_noise_blockdiag: bad argument");
    return launch_noise_blockdiag(Ls, mu, eps, 0u, 0u, 0, N, a_out, (hipStream_t)stream, nullptr, covo_propagate_nan(h));
}

int covo_noise_blockdiag_philox(covo_handle_t h, const float *Ls, const float *mu, uint32_t key0, uint32_t key1,
                                int64_t sample_offset, int32_t N, float *a_out, void *stream)
{
    REQUIRE(h, "covo_noise_blockdiag_philox: null handle");
    REQUIRE(Ls && mu && a_out && N > 0, "covo_noise_blockdiag_philox: bad argument");
    return launch_noise_blockdiag(Ls, mu, nullptr, key0, key1, sample_offset, N, a_out, (hipStream_t)stream, nullptr,
                                  covo_propagate_nan(h));
}

int covo_rollout_cost(covo_handle_t h, const float *state, const float *pos_traj, const float *vel_traj, int32_t T,
                      const covo_env_params *params, const float *f_disturb_shared, const float *f_disturb_steps,
                      const float *a, int32_t N, float *cost_out, float *groupmin, double *pos_stats, void *stream)
{
    REQUIRE(h, "covo_rollout_cost: null handle");
    CHECK_DEVICE(h, "covo_rollout_cost");
    REQUIRE(state && pos_traj && vel_traj && params && a && cost_out && T > 0, "covo_rollout_cost: bad argument");
    REQUIRE(N > 0 && N <= h->cfg.n_local, "covo_rollout_cost: N=%d outside (0, n_local=%d]", N, h->cfg.n_local);
    CHECK_MODEL(params, "covo_rollout_cost");
    REQUIRE(!needs_table(params) || f_disturb_steps, "covo_rollout_cost: disturb_kind=%d needs f_disturb_steps (covo_disturb_table)",
            params->disturb_kind);
    return launch_rollout(state, pos_traj, vel_traj, T, *params, f_disturb_shared, a, N, h->cfg.discount,
                          (h->cfg.flags & COVO_FLAG_ACTIONS_CLIPPED) != 0, cost_out, groupmin, pos_stats, h->ws_stats,
                          (hipStream_t)stream, nullptr, nullptr, 0.0f, f_disturb_steps, 0, covo_propagate_nan(h));
}

int covo_disturb_table(covo_handle_t h, const covo_env_params *params, const float *state, int32_t batch,
                       const uint32_t *keys_dev, uint32_t key0, uint32_t key1, int32_t key_mode, int32_t deterministic,
                       float *out, void *stream)
{
    REQUIRE(h, "covo_disturb_table: null handle");
    REQUIRE(params && state && out && batch > 0, "covo_disturb_table: bad argument");
    REQUIRE(key_mode >= COVO_DISTURB_KEYS_SHARED && key_mode <= COVO_DISTURB_KEYS_NOMINAL, "covo_disturb_table: key_mode=%d", key_mode);
    CHECK_MODEL(params, "covo_disturb_table");
    return launch_disturb_table(*params, state, batch, keys_dev, key0, key1, key_mode, deterministic, out, (hipStream_t)stream);
}

__global__ void pos_info_kernel(const double *__restrict__ stats, const float *__restrict__ state, double inv_n,
                                float *__restrict__ mean, float *__restrict__ sd)
{
    const int t = threadIdx.x;  // (step, axis)
    if (t >= COVO_H * 3) return;
    const int k = t / 3, ax = t % 3;
    const double m1 = stats[k * 6 + ax] * inv_n;
    const double var = fmax(stats[k * 6 + 3 + ax] * inv_n - m1 * m1, 0.0);
    mean[t] = (float)((double)state[ST_POS + ax] + m1);
    sd[t] = (float)sqrt(var);
}

int covo_pos_info(covo_handle_t h, const double *pos_stats, const float *state, int64_t n_total, float *pos_mean_out,
                  float *pos_std_out, void *stream)
{
    REQUIRE(h, "covo_pos_info: null handle");
    CHECK_DEVICE(h, "covo_pos_info");
    REQUIRE(pos_stats && state && pos_mean_out && pos_std_out && n_total > 0, "covo_pos_info: bad argument");
    hipLaunchKernelGGL(pos_info_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, pos_stats, state, 1.0 / (double)n_total,
                       pos_mean_out, pos_std_out);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}

int covo_debug_time_rollout(covo_handle_t h, const float *state, const float *pos_traj, const float *vel_traj, int32_t T,
                            const covo_env_params *params, const float *f_disturb_shared, const float *f_disturb_steps,
                            const float *a, int32_t N, float *cost_out, float *groupmin, int32_t with_records, int32_t reps,
                            float *us_out, void *stream)
{
    REQUIRE(h, "covo_debug_time_rollout: null handle");
    REQUIRE(state && pos_traj && vel_traj && params && a && cost_out && us_out && T > 0 && reps > 0, "covo_debug_time_rollout: bad argument");
    REQUIRE(N > 0 && N <= h->cfg.n_local, "covo_debug_time_rollout: N=%d outside (0, n_local=%d]", N, h->cfg.n_local);
    CHECK_MODEL(params, "covo_debug_time_rollout");
    REQUIRE(!needs_table(params) || f_disturb_steps, "covo_debug_time_rollout: disturb_kind=%d needs f_disturb_steps", params->disturb_kind);
    hipStream_t s = (hipStream_t)stream;
    const bool clipped = (h->cfg.flags & COVO_FLAG_ACTIONS_CLIPPED) != 0;
    // with_records: the variant the fused step runs (every workgroup also leaves its online-softmax record), when the launch
    // shape allows it there (step.hip: enqueue_step)
    const bool rec = with_records && rollout_workgroups(N, false) <= h->max_red_blocks;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t err = hipEventCreate(&e0);
    if (err == hipSuccess) err = hipEventCreate(&e1);
    int rc = 0;
    auto launch = [&]() {
        return launch_rollout(state, pos_traj, vel_traj, T, *params, f_disturb_shared, a, N, h->cfg.discount, clipped, cost_out,
                              rec ? nullptr : groupmin, nullptr, h->ws_stats, s, nullptr, rec ? h->ws_partials : nullptr, h->cfg.lam,
                              f_disturb_steps);
    };
    for (int i = 0; i < 3 && !rc && err == hipSuccess; ++i) rc = launch();
    constexpr int BATCHES = 3;
    float best = 1e30f, sum = 0.0f;
    for (int it = 0; it < BATCHES && !rc && err == hipSuccess; ++it) {
        err = hipEventRecord(e0, s);
        for (int i = 0; i < reps && !rc; ++i) rc = launch();
        if (err == hipSuccess) err = hipEventRecord(e1, s);
        if (err == hipSuccess) err = hipStreamSynchronize(s);
        float ms = 0.0f;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        sum += ms;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (err != hipSuccess) {
        covo_set_error("covo_debug_time_rollout: %s", hipGetErrorString(err));
        return (int)err;
    }
    us_out[0] = sum * 1e3f / (float)(reps * BATCHES);  // mean over all launches
    us_out[1] = best * 1e3f / (float)reps;             // the fastest batch of `reps`
    return 0;
}

int covo_softmax_reduce(covo_handle_t h, const float *cost, const float *a, int32_t N, const float *groupmin,
                        float *partial_out, void *stream)
{
    REQUIRE(h, "covo_softmax_reduce: null handle");
    CHECK_DEVICE(h, "covo_softmax_reduce");
    REQUIRE(cost && a && partial_out, "covo_softmax_reduce: bad argument");
    REQUIRE(N > 0 && N <= h->cfg.n_local, "covo_softmax_reduce: N=%d outside (0, n_local=%d]", N, h->cfg.n_local);
    return launch_softmax_reduce(h, cost, a, N, groupmin, (N + 63) / 64, partial_out, nullptr, 1.0f, nullptr,
                                 (hipStream_t)stream);
}

int covo_softmax_update(covo_handle_t h, const float *cost, const float *a, int32_t N, const float *groupmin,
                        const float *a_mean_old, float gamma_mean, float *a_mean_out, void *stream)
{
    REQUIRE(h, "covo_softmax_update: null handle");
    CHECK_DEVICE(h, "covo_softmax_update");
    REQUIRE(cost && a && a_mean_old && a_mean_out, "covo_softmax_update: bad argument");
    REQUIRE(N > 0 && N <= h->cfg.n_local, "covo_softmax_update: N=%d outside (0, n_local=%d]", N, h->cfg.n_local);
    return launch_softmax_reduce(h, cost, a, N, groupmin, (N + 63) / 64, nullptr, a_mean_old, gamma_mean, a_mean_out,
                                 (hipStream_t)stream);
}

int covo_softmax_update_cov(covo_handle_t h, const float *cost, const float *a, int32_t N, const float *groupmin,
                            const float *a_mean_old, float gamma_mean, const float *a_cov_old, float gamma_sigma,
                            float *a_mean_out, float *a_cov_out, void *stream)
{
    REQUIRE(h, "covo_softmax_update_cov: null handle");
    CHECK_DEVICE(h, "covo_softmax_update_cov");
    REQUIRE(cost && a && a_mean_old && a_cov_old && a_mean_out && a_cov_out, "covo_softmax_update_cov: bad argument");
    REQUIRE(N > 0 && N <= h->cfg.n_local, "covo_softmax_update_cov: N=%d outside (0, n_local=%d]", N, h->cfg.n_local);
    return launch_softmax_update_cov(h, cost, a, N, groupmin, (N + 63) / 64, a_mean_old, gamma_mean, a_cov_old, gamma_sigma,
                                     a_mean_out, a_cov_out, (hipStream_t)stream);
}

int covo_softmax_reduce_cov(covo_handle_t h, const float *cost, const float *a, int32_t N, const float *groupmin,
                            const float *a_mean_old, float *record_out, void *stream)
{
    REQUIRE(h, "covo_softmax_reduce_cov: null handle");
    CHECK_DEVICE(h, "covo_softmax_reduce_cov");
    REQUIRE(cost && a && a_mean_old && record_out, "covo_softmax_reduce_cov: bad argument");
    REQUIRE(N > 0 && N <= h->cfg.n_local, "covo_softmax_reduce_cov: N=%d outside (0, n_local=%d]", N, h->cfg.n_local);
    return launch_softmax_reduce_cov(h, cost, a, N, groupmin, (N + 63) / 64, a_mean_old, record_out, (hipStream_t)stream);
}

int covo_merge_ranks_cov(covo_handle_t h, const float *records, int32_t G, const float *a_mean_old, float gamma_mean,
                         const float *a_cov_old, float gamma_sigma, float *a_mean_out, float *a_cov_out, double *pos_stats_out,
                         void *stream)
{
    REQUIRE(h, "covo_merge_ranks_cov: null handle");
    CHECK_DEVICE(h, "covo_merge_ranks_cov");
    REQUIRE(records && a_mean_old && a_cov_old && a_mean_out && a_cov_out && G > 0, "covo_merge_ranks_cov: bad argument");
    int rc = launch_merge_cov(records, G, COVO_RANK_RECORD_COV_FLOATS, h->cfg.lam, a_mean_old, gamma_mean, a_cov_old, gamma_sigma,
                              a_mean_out, a_cov_out, (hipStream_t)stream);
    if (rc) return rc;
    if (pos_stats_out != nullptr) rc = launch_rank_stats_sum(records, G, pos_stats_out, (hipStream_t)stream, true);
    return rc;
}

int covo_merge(covo_handle_t h, const float *partials, int32_t G, const float *a_mean_old, float gamma_mean,
               float *a_mean_out, void *stream)
{
    REQUIRE(h, "covo_merge: null handle");
    CHECK_DEVICE(h, "covo_merge");
    REQUIRE(partials && a_mean_old && a_mean_out && G > 0, "covo_merge: bad argument");
    return launch_merge(partials, G, h->cfg.lam, a_mean_old, gamma_mean, a_mean_out, (hipStream_t)stream);
}

int covo_merge_ranks(covo_handle_t h, const float *records, int32_t G, const float *a_mean_old, float gamma_mean,
                     float *a_mean_out, double *pos_stats_out, void *stream)
{
    REQUIRE(h, "covo_merge_ranks: null handle");
    CHECK_DEVICE(h, "covo_merge_ranks");
    REQUIRE(records && a_mean_old && a_mean_out && G > 0, "covo_merge_ranks: bad argument");
    int rc = launch_merge(records, G, h->cfg.lam, a_mean_old, gamma_mean, a_mean_out, (hipStream_t)stream, nullptr, 1,
                          COVO_RANK_RECORD_FLOATS);
    if (rc) return rc;
    if (pos_stats_out != nullptr) rc = launch_rank_stats_sum(records, G, pos_stats_out, (hipStream_t)stream);
    return rc;
}

int covo_merge_ranks_wide(covo_handle_t h, const float *records, int32_t G, int32_t record_floats, const float *a_mean_old,
                          float gamma_mean, float *a_mean_out, double *pos_stats_out, void *stream)
{
    REQUIRE(h, "covo_merge_ranks_wide: null handle");
    CHECK_DEVICE(h, "covo_merge_ranks_wide");
    REQUIRE(records && a_mean_old && a_mean_out && G > 0, "covo_merge_ranks_wide: bad argument");
    REQUIRE(record_floats == COVO_RANK_RECORD_FLOATS || record_floats == COVO_RANK_RECORD_COV_FLOATS,
            "covo_merge_ranks_wide: record_floats=%d is neither COVO_RANK_RECORD_FLOATS nor COVO_RANK_RECORD_COV_FLOATS", record_floats);
    int rc = launch_merge(records, G, h->cfg.lam, a_mean_old, gamma_mean, a_mean_out, (hipStream_t)stream, nullptr, 1, record_floats);
    if (rc) return rc;
    if (pos_stats_out != nullptr)
        rc = launch_rank_stats_sum(records, G, pos_stats_out, (hipStream_t)stream, record_floats == COVO_RANK_RECORD_COV_FLOATS);
    return rc;
}

int covo_exchange_create(covo_handle_t h, int32_t world, int32_t rank, void *handle_out)
{
    REQUIRE(h && handle_out, "covo_exchange_create: null argument");
    return exchange_create(h, world, rank, handle_out);
}

int covo_exchange_connect(covo_handle_t h, const void *handles)
{
    REQUIRE(h && handles, "covo_exchange_connect: null argument");
    return exchange_connect(h, handles);
}

int covo_exchange_set_timeout(covo_handle_t h, double seconds)
{
    REQUIRE(h, "covo_exchange_set_timeout: null handle");
    return exchange_set_timeout(h, seconds);
}

int covo_device_bus_id(int32_t device, char *out, int32_t len)
{
    REQUIRE(out && len >= 16, "covo_device_bus_id: out == NULL or len < 16");
    COVO_CHECK_HIP(hipDeviceGetPCIBusId(out, len, device));
    return 0;
}

int covo_exchange_records(covo_handle_t h, const float *record, float *gathered_out, void *stream)
{
    REQUIRE(h && record && gathered_out, "covo_exchange_records: null argument");
    CHECK_DEVICE(h, "covo_exchange_records");
    return exchange_records(h, record, gathered_out, nullptr, (hipStream_t)stream);
}

int covo_exchange_records_cov(covo_handle_t h, const float *record, float *gathered_out, void *stream)
{
    REQUIRE(h && record && gathered_out, "covo_exchange_records_cov: null argument");
    CHECK_DEVICE(h, "covo_exchange_records_cov");
    return exchange_records(h, record, gathered_out, nullptr, (hipStream_t)stream, COVO_RANK_RECORD_COV_FLOATS);
}

int covo_shift_mean(covo_handle_t h, const float *a_mean_in, float *a_mean_out, void *stream)
{
    REQUIRE(h, "covo_shift_mean: null handle");
    REQUIRE(a_mean_in && a_mean_out && a_mean_in != a_mean_out, "covo_shift_mean: bad argument (in must differ from out)");
    return launch_shift_mean(a_mean_in, a_mean_out, (hipStream_t)stream);
}

int covo_hessian(covo_handle_t h, const float *state, const float *pos_traj, const float *vel_traj, int32_t T,
                 const covo_env_params *params, const float *a_mean, const float *f_disturb_steps, int32_t batch,
                 double *R_out, void *stream)
{
    REQUIRE(h, "covo_hessian: null handle");
    CHECK_DEVICE(h, "covo_hessian");
    REQUIRE(state && pos_traj && vel_traj && params && a_mean && R_out && T > 0 && batch > 0, "covo_hessian: bad argument");
    CHECK_MODEL(params, "covo_hessian");
    REQUIRE(!needs_table(params) || f_disturb_steps, "covo_hessian: disturb_kind=%d needs f_disturb_steps (covo_disturb_table)",
            params->disturb_kind);
    const size_t need = hessian_workspace_bytes(batch);
    if (need > h->ws_hess_bytes) {  // only for batch sizes not seen before (never inside the steady-state step)
        COVO_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
        step_graphs_drop(h);  // captured step graphs hold the old address
        (void)hipFree(h->ws_hess);
        h->ws_hess = nullptr;
        h->ws_hess_bytes = 0;
        COVO_CHECK_HIP(hipMalloc(&h->ws_hess, need));
        h->ws_hess_bytes = need;
    }
    return launch_hessian(state, pos_traj, vel_traj, T, *params, a_mean, batch, R_out, h->ws_hess, (hipStream_t)stream, nullptr, 0,
                          nullptr, f_disturb_steps, nullptr, h->status_dev);
}

int covo_hessian_pairs(covo_handle_t h, const float *state, const float *pos_traj, const float *vel_traj, int32_t T,
                       const covo_env_params *params, const float *a_mean, const float *f_disturb_steps, int32_t batch,
                       double *R_out, void *stream)
{
    REQUIRE(h, "covo_hessian_pairs: null handle");
    REQUIRE(state && pos_traj && vel_traj && params && a_mean && R_out && T > 0 && batch > 0, "covo_hessian_pairs: bad argument");
    CHECK_MODEL(params, "covo_hessian_pairs");
    return launch_hessian_pairs(state, pos_traj, vel_traj, T, *params, a_mean, batch, R_out, (hipStream_t)stream, f_disturb_steps);
}

int covo_sigma(covo_handle_t h, const double *R, int32_t batch, float sample_sigma, float *Sigma_out, float *L_out,
               void *stream)
{
    REQUIRE(h, "covo_sigma: null handle");
    CHECK_DEVICE(h, "covo_sigma");
    REQUIRE(R && L_out && batch > 0 && sample_sigma > 0.0f, "covo_sigma: bad argument");
    const size_t need = sigma_ns_workspace_bytes(batch);
    if (need > h->ws_sigma_bytes) {  // only for batch sizes not seen before (never inside the steady-state step)
        COVO_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
        step_graphs_drop(h);  // captured step graphs hold the old address
        (void)hipFree(h->ws_sigma);
        h->ws_sigma = nullptr;
        h->ws_sigma_bytes = 0;
        COVO_CHECK_HIP(hipMalloc(&h->ws_sigma, need));
        h->ws_sigma_bytes = need;
    }
    return launch_sigma_ns(h->opt, R, batch, sample_sigma, Sigma_out, L_out, h->ws_sigma, (hipStream_t)stream, nullptr, h->status_dev,
                           (h->cfg.flags & COVO_FLAG_SHARED_DEVICE) == 0);
}

// ---- per-HANDLE experiment switches (CovoOpts, covo_common.hpp); every setter bumps the handle's epoch: its captured step graphs
// hold the old launch set / kernel arguments and are re-captured at the next step
int covo_debug_set_ns_tail(covo_handle_t h, int n_squarings, int n_iters)
{
    REQUIRE(h, "covo_debug_set_ns_tail: null handle");
    if (n_squarings > 64 || n_iters > 64 || (n_squarings < 0) != (n_iters < 0)) {
        covo_set_error("covo_debug_set_ns_tail: (%d, %d) out of range", n_squarings, n_iters);
        return COVO_E_BADARG;
    }
    if (n_squarings < 0) {  // back to the defaults, defined in ONE place (sigma_ns.hip)
        sigma_ns_tail_defaults(h->opt);
    } else {
        h->opt.ns_tail_squarings = h->opt.ns_tail_squarings_batched = n_squarings;  // batch 1 and batched launches alike
        h->opt.ns_tail_iters = h->opt.ns_tail_iters_batched = n_iters;
    }
    ++h->opt.epoch;
    return 0;
}

int covo_debug_set_stream_gemm(covo_handle_t h, int on)
{
    REQUIRE(h, "covo_debug_set_stream_gemm: null handle");
    h->opt.stream_gemm = on ? 1 : 0;
    ++h->opt.epoch;
    return 0;
}

int covo_debug_set_fuse_small(covo_handle_t h, int on)
{
    REQUIRE(h, "covo_debug_set_fuse_small: null handle");
    h->opt.fuse_small = on ? 1 : 0;
    ++h->opt.epoch;
    return 0;
}

int covo_debug_set_fold_begin(covo_handle_t h, int on)
{
    REQUIRE(h, "covo_debug_set_fold_begin: null handle");
    h->opt.fold_begin = on ? 1 : 0;
    ++h->opt.epoch;
    return 0;
}

int covo_debug_set_ns_ritz_inside(covo_handle_t h, int on)
{
    REQUIRE(h, "covo_debug_set_ns_ritz_inside: null handle");
    h->opt.ns_ritz_inside = (on == 2) ? 2 : (on ? 1 : 0);  // (2: timing reference, the last iterate only)
    ++h->opt.epoch;
    return 0;
}

int covo_debug_set_ns_deflate(covo_handle_t h, int on)
{
    REQUIRE(h, "covo_debug_set_ns_deflate: null handle");
    h->opt.ns_deflate = on ? 1 : 0;
    ++h->opt.epoch;
    return 0;
}

int covo_debug_set_ns_merged(covo_handle_t h, int on)
{
    REQUIRE(h, "covo_debug_set_ns_merged: null handle");
    h->opt.ns_merged = on ? 1 : 0;
    ++h->opt.epoch;
    return 0;
}

int covo_debug_set_ns_coherence(covo_handle_t h, int force_agent)
{
    REQUIRE(h, "covo_debug_set_ns_coherence: null handle");
    h->opt.ns_force_agent = force_agent ? 1 : 0;
    ++h->opt.epoch;
    return 0;
}

int covo_debug_sigma_workspace(covo_handle_t h, double *out, int64_t offset_doubles, int64_t count, void *stream)
{
    REQUIRE(h && out, "covo_debug_sigma_workspace: bad argument");
    REQUIRE((size_t)(offset_doubles + count) * sizeof(double) <= h->ws_sigma_bytes, "covo_debug_sigma_workspace: range");
    COVO_CHECK_HIP(hipMemcpyAsync(out, (const double *)h->ws_sigma + offset_doubles, (size_t)count * sizeof(double),
                                  hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int covo_env_step(covo_handle_t h, float *state, float *noisy_state, const float *pos_traj, const float *vel_traj,
                  const float *acc_traj, int32_t T, const covo_env_params *params, const float *action,
                  const uint32_t *step_key, int32_t noisy_on, float obs_noise_scale, float *log, int32_t log_index,
                  void *stream)
{
    REQUIRE(h, "covo_env_step: null handle");
    REQUIRE(state && noisy_state && pos_traj && vel_traj && acc_traj && params && action && step_key && T > 0 &&
                log_index >= 0,
            "covo_env_step: bad argument");
    CHECK_MODEL(params, "covo_env_step");
    return launch_env_step(state, noisy_state, pos_traj, vel_traj, acc_traj, T, *params, action, step_key, noisy_on,
                           obs_noise_scale, log, log_index, (hipStream_t)stream);
}

int covo_pid_nominal(covo_handle_t h, const float *state0, const float *pos_traj, const float *vel_traj,
                     const float *acc_traj, int32_t T, const covo_env_params *params, const covo_env_params *pid_params,
                     float Kp, float Kd, float Kp_att, uint32_t key0, uint32_t key1, int32_t n_steps,
                     float *states_out, float *a_means_out, uint32_t *keys_out, void *stream)
{
    REQUIRE(h, "covo_pid_nominal: null handle");
    REQUIRE(state0 && pos_traj && vel_traj && acc_traj && params && pid_params && states_out && a_means_out && T > 0 &&
                n_steps > 0,
            "covo_pid_nominal: bad argument");
    CHECK_MODEL(params, "covo_pid_nominal");
    return launch_pid_nominal(state0, pos_traj, vel_traj, acc_traj, T, *params, *pid_params, Kp, Kd, Kp_att, key0, key1, n_steps,
                              states_out, a_means_out, keys_out, (hipStream_t)stream);
}

// Philox4x32-10 on the host: child i of split(key, num) as covo_mpc_amd/random.py forms it
static void host_philox_split(const uint32_t key[2], uint32_t i, uint32_t child[2])
{
    uint32_t c0 = i, c1 = 0, c2 = 0, c3 = 0x5EEDu, k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    child[0] = c0;
    child[1] = c1;
}

int covo_run_episode(covo_handle_t h, const covo_env_params *params, const covo_step_args *args, float *state_true,
                     const float *acc_traj, int32_t noisy_on, float obs_noise_scale, float *log, uint32_t *rng,
                     int32_t n_steps, void *stream)
{
    REQUIRE(h, "covo_run_episode: null handle");
    CHECK_DEVICE(h, "covo_run_episode");
    REQUIRE(params && args && state_true && acc_traj && rng && n_steps > 0, "covo_run_episode: bad argument");
    REQUIRE(args->derive_keys == 1, "covo_run_episode: args->derive_keys must be 1 (the controller key is the raw rng_act)");
    REQUIRE(args->a_mean_in == nullptr, "covo_run_episode: args->a_mean_in must be NULL (the episode carries the mean in args->a_mean)");
    CHECK_MODEL(params, "covo_run_episode");
    REQUIRE(args->partial_out == nullptr || (exchange_ready(h) && args->a_mean_shift != nullptr),
            "covo_run_episode: a sample-sharded step (partial_out != NULL) needs the peer-write exchange (covo_exchange_create / "
            "covo_exchange_connect) and a_mean_shift");
    REQUIRE(args->state && args->pos_traj && args->vel_traj && args->a_mean && args->a && args->cost && args->groupmin &&
                args->T > 0 && args->mode >= 0 && args->mode <= 2,
            "covo_run_episode: bad step arguments");
    hipStream_t s = (hipStream_t)stream;
    uint32_t key[2] = {rng[0], rng[1]};
    for (int t = 0; t < n_steps; ++t) {
        // run_one_step (quadrotor.py:520-538): rng, rng_act, rng_step, rng_control = split(rng, 4); ...; rng, _ = split(rng)
        uint32_t nrng[2], rng_act[2], rng_step[2];
        host_philox_split(key, 0u, nrng);
        host_philox_split(key, 1u, rng_act);
        host_philox_split(key, 2u, rng_step);
        int rc = covo_step_impl(h, params, args, rng_act[0], rng_act[1], nullptr, s);
        if (rc) return rc;
        if (args->partial_out != nullptr) {
            // sample-sharded: every rank's record to every rank (peer writes, exchange.hip), then the same merge on all of them.
            // partial_out is this rank's RANK record: {m, s, v} there, its position sums (if any) at + COVO_PARTIAL_FLOATS
            const float *gathered = nullptr;
            if (args->mode == COVO_MODE_MPPI && args->gamma_sigma != 0.0f) {
                // mppi.py:119-125 on sharded ranks: the record also carries the 320 second moments (COVO_RANK_RECORD_COV_FLOATS);
                // a_cov was shifted in place by the begin launch and is adapted in place, identically on every rank
                if ((rc = exchange_records(h, args->partial_out, nullptr, &gathered, s, COVO_RANK_RECORD_COV_FLOATS))) return rc;
                if ((rc = launch_merge_cov(gathered, exchange_world(h), COVO_RANK_RECORD_COV_FLOATS, h->cfg.lam, args->a_mean_shift,
                                           args->gamma_mean, args->a_cov, args->gamma_sigma, args->a_mean, args->a_cov, s)))
                    return rc;
            } else {
                if ((rc = exchange_records(h, args->partial_out, nullptr, &gathered, s))) return rc;
                if ((rc = launch_merge(gathered, exchange_world(h), h->cfg.lam, args->a_mean_shift, args->gamma_mean, args->a_mean, s,
                                       nullptr, 1, COVO_RANK_RECORD_FLOATS)))
                    return rc;
            }
        }
        rc = launch_env_step(state_true, const_cast<float *>(args->state), args->pos_traj, args->vel_traj, acc_traj, args->T,
                             *params, args->a_mean, rng_step, noisy_on, obs_noise_scale, log, t, s);
        if (rc) return rc;
        host_philox_split(nrng, 0u, key);
    }
    rng[0] = key[0];
    rng[1] = key[1];
    return 0;
}

static int check_batch_models(const covo_env_params *params, int E, const char *what)
{
    for (int e = 0; e < E; ++e) {
        CHECK_MODEL(&params[e], what);
        if (params[e].reward_kind != params[0].reward_kind || params[e].rollover_terminate != params[0].rollover_terminate ||
            params[e].disturb_kind != params[0].disturb_kind || params[e].max_steps_in_episode != params[0].max_steps_in_episode ||
            params[e].reset_traj != params[0].reset_traj) {
            covo_set_error("%s: instance %d differs from instance 0 in reward_kind / rollover_terminate / disturb_kind / "
                           "max_steps_in_episode / reset_traj (one kernel variant per launch)", what, e);
            return COVO_E_BADARG;
        }
    }
    return 0;
}

int covo_env_step_batched(covo_handle_t h, int32_t n_envs, float *states, float *noisy_states, const float *pos_traj,
                          const float *vel_traj, const float *acc_traj, int32_t T, const covo_env_params *params,
                          const float *a_mean, const uint32_t *step_keys, int32_t noisy_on, float obs_noise_scale, float *log,
                          int32_t log_stride, int32_t log_index, void *stream)
{
    REQUIRE(h, "covo_env_step_batched: null handle");
    CHECK_DEVICE(h, "covo_env_step_batched");
    REQUIRE(n_envs > 0 && n_envs <= COVO_MAX_ENVS, "covo_env_step_batched: n_envs=%d outside (0, %d]", n_envs, COVO_MAX_ENVS);
    REQUIRE(states && noisy_states && pos_traj && vel_traj && acc_traj && params && a_mean && step_keys && T > 0,
            "covo_env_step_batched: bad argument");
    REQUIRE(log == nullptr || (log_index >= 0 && log_index < log_stride), "covo_env_step_batched: log_index=%d outside [0, %d)",
            log_index, log_stride);
    int rc = check_batch_models(params, n_envs, "covo_env_step_batched");
    if (rc) return rc;
    const void *inst = nullptr;
    if ((rc = batch_env_inst(h, params, n_envs, (hipStream_t)stream, &inst))) return rc;
    return launch_env_step_batched(states, noisy_states, pos_traj, vel_traj, acc_traj, T, params[0], inst, n_envs, a_mean, step_keys,
                                   noisy_on, obs_noise_scale, log, log_stride, log_index, (hipStream_t)stream);
}

int covo_run_episode_batched(covo_handle_t h, const covo_batch_args *args, const covo_env_params *params, float *states_true,
                             const float *acc_traj, int32_t noisy_on, float obs_noise_scale, float *log, int32_t log_stride,
                             int32_t log_index, uint32_t *rngs, int32_t n_steps, void *stream)
{
    REQUIRE(h, "covo_run_episode_batched: null handle");
    CHECK_DEVICE(h, "covo_run_episode_batched");
    REQUIRE(args && params && states_true && acc_traj && rngs && n_steps > 0, "covo_run_episode_batched: bad argument");
    const int E = args->n_envs;
    REQUIRE(E > 0 && E <= COVO_MAX_ENVS, "covo_run_episode_batched: n_envs=%d outside (0, %d]", E, COVO_MAX_ENVS);
    REQUIRE(args->n_samples > 0 && args->n_samples <= h->cfg.n_local, "covo_run_episode_batched: n_samples=%d outside (0, %d]",
            args->n_samples, h->cfg.n_local);
    REQUIRE(args->states && args->pos_traj && args->vel_traj && args->a_mean && args->a && args->cost && args->groupmin && args->T > 0,
            "covo_run_episode_batched: bad step arguments");
    REQUIRE(log == nullptr || (log_index >= 0 && log_index + n_steps <= log_stride),
            "covo_run_episode_batched: log rows [%d, %d) outside [0, %d)", log_index, log_index + n_steps, log_stride);
    int rc = check_batch_models(params, E, "covo_run_episode_batched");
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const void *inst = nullptr;
    if ((rc = batch_env_inst(h, params, E, s, &inst))) return rc;
    uint32_t act_keys[2 * COVO_MAX_ENVS], step_keys[2 * COVO_MAX_ENVS], next[2 * COVO_MAX_ENVS];
    for (int t = 0; t < n_steps; ++t) {
        // per instance, run_one_step (quadrotor.py:520-538): rng, rng_act, rng_step, rng_control = split(rng, 4); ...; rng, _ = split(rng)
        for (int e = 0; e < E; ++e) {
            const uint32_t key[2] = {rngs[2 * e], rngs[2 * e + 1]};
            uint32_t nrng[2];
            host_philox_split(key, 0u, nrng);
            host_philox_split(key, 1u, &act_keys[2 * e]);
            host_philox_split(key, 2u, &step_keys[2 * e]);
            host_philox_split(nrng, 0u, &next[2 * e]);
        }
        if ((rc = covo_step_batched_impl(h, args, params, act_keys, s))) return rc;
        if ((rc = launch_env_step_batched(states_true, const_cast<float *>(args->states), args->pos_traj, args->vel_traj, acc_traj,
                                          args->T, params[0], inst, E, args->a_mean, step_keys, noisy_on, obs_noise_scale, log,
                                          log_stride, log_index + t, s)))
            return rc;
        std::memcpy(rngs, next, (size_t)2 * E * sizeof(uint32_t));
    }
    return 0;
}

int covo_debug_time_batched(covo_handle_t h, int32_t step_mask, int32_t reps, float *us_out, void *stream)
{
    REQUIRE(h && us_out && reps > 0, "covo_debug_time_batched: bad argument");
    CHECK_DEVICE(h, "covo_debug_time_batched");
    return covo_debug_time_batched_impl(h, step_mask, reps, us_out, (hipStream_t)stream);
}

int covo_mpc_step_batched(covo_handle_t h, const covo_batch_args *args, const covo_env_params *params, const uint32_t *keys,
                          void *stream)
{
    REQUIRE(h, "covo_mpc_step_batched: null handle");
    CHECK_DEVICE(h, "covo_mpc_step_batched");
    REQUIRE(args && params && keys, "covo_mpc_step_batched: null argument");
    REQUIRE(args->n_envs > 0 && args->n_envs <= COVO_MAX_ENVS, "covo_mpc_step_batched: n_envs=%d outside (0, %d]", args->n_envs,
            COVO_MAX_ENVS);
    REQUIRE(args->n_samples > 0 && args->n_samples <= h->cfg.n_local, "covo_mpc_step_batched: n_samples=%d outside (0, %d]",
            args->n_samples, h->cfg.n_local);
    REQUIRE(args->states && args->pos_traj && args->vel_traj && args->a_mean && args->a && args->cost && args->groupmin &&
                args->T > 0,
            "covo_mpc_step_batched: null buffer");
    for (int e = 0; e < args->n_envs; ++e) {
        CHECK_MODEL(&params[e], "covo_mpc_step_batched");
        REQUIRE(params[e].reward_kind == params[0].reward_kind && params[e].rollover_terminate == params[0].rollover_terminate &&
                    params[e].disturb_kind == params[0].disturb_kind,
                "covo_mpc_step_batched: all instances must share reward_kind, rollover_terminate and disturb_kind (one kernel "
                "variant per launch); disturb_params / period / scale may differ");
    }
    return covo_step_batched_impl(h, args, params, keys, (hipStream_t)stream);
}

int covo_debug_time_step(covo_handle_t h, const covo_env_params *params, const covo_step_args *args, int32_t step_mask,
                         int32_t hess_mask, int32_t sigma_stages, int32_t reps, float *us_out, void *stream)
{
    REQUIRE(h && params && args && us_out && reps > 0, "covo_debug_time_step: bad argument");
    return covo_debug_time_step_impl(h, params, args, step_mask, hess_mask, sigma_stages, reps, us_out, (hipStream_t)stream);
}

int covo_debug_hess_workspace(covo_handle_t h, double *out, int64_t offset_doubles, int64_t count, void *stream)
{
    REQUIRE(h && out, "covo_debug_hess_workspace: bad argument");
    REQUIRE((size_t)(offset_doubles + count) * sizeof(double) <= h->ws_hess_bytes, "covo_debug_hess_workspace: range");
    COVO_CHECK_HIP(hipMemcpyAsync(out, (const double *)h->ws_hess + offset_doubles, (size_t)count * sizeof(double),
                                  hipMemcpyDeviceToHost, (hipStream_t)stream));
    return 0;
}

int covo_debug_batched_hessians(covo_handle_t h, double *out, int64_t offset_doubles, int64_t count, void *stream)
{
    REQUIRE(h && out && offset_doubles >= 0 && count > 0, "covo_debug_batched_hessians: bad argument");
    return covo_debug_batched_hessians_impl(h, out, offset_doubles, count, (hipStream_t)stream);
}

int covo_sigma_jacobi(covo_handle_t h, const double *R, int32_t batch, float sample_sigma, float *Sigma_out, float *L_out,
                      void *stream)
{
    REQUIRE(h, "covo_sigma_jacobi: null handle");
    REQUIRE(R && L_out && batch > 0 && sample_sigma > 0.0f, "covo_sigma_jacobi: bad argument");
    return launch_sigma(R, batch, sample_sigma, Sigma_out, L_out, nullptr, (hipStream_t)stream);
}

int covo_sigma_profile(covo_handle_t h, const double *R, float sample_sigma, float *Sigma_out, float *L_out,
                       uint64_t *ticks_out, void *stream)
{
    REQUIRE(h, "covo_sigma_profile: null handle");
    REQUIRE(R && L_out && ticks_out, "covo_sigma_profile: bad argument");
    return launch_sigma(R, 1, sample_sigma, Sigma_out, L_out, (unsigned long long *)ticks_out, (hipStream_t)stream);
}

int covo_mpc_step(covo_handle_t h, const covo_env_params *params, const covo_step_args *args, uint32_t key0, uint32_t key1,
                  const float *f_disturb_shared, void *stream)
{
    REQUIRE(h, "covo_mpc_step: null handle");
    CHECK_DEVICE(h, "covo_mpc_step");
    REQUIRE(params && args, "covo_mpc_step: null argument");
    REQUIRE(args->mode >= 0 && args->mode <= 2, "covo_mpc_step: mode=%d", args->mode);
    REQUIRE(args->n_samples > 0 && args->n_samples <= h->cfg.n_local, "covo_mpc_step: n_samples=%d outside (0, %d]",
            args->n_samples, h->cfg.n_local);
    REQUIRE(args->state && args->pos_traj && args->vel_traj && args->a_mean && args->a && args->cost && args->groupmin &&
                args->T > 0,
            "covo_mpc_step: null buffer");
    REQUIRE(args->mode != COVO_MODE_COVO_OFFLINE || (args->L_table && args->n_table > 0), "covo_mpc_step: offline needs L_table");
    REQUIRE(args->mode != COVO_MODE_MPPI || args->a_cov, "covo_mpc_step: mppi needs a_cov");
    CHECK_MODEL(params, "covo_mpc_step");
    REQUIRE(args->gamma_sigma == 0.0f || args->mode == COVO_MODE_MPPI,
            "covo_mpc_step: gamma_sigma != 0 is MPPI's covariance adaptation (mppi.py:119-125)");
    REQUIRE(!needs_table(params) || args->derive_keys == 1, "covo_mpc_step: disturb_kind=%d needs derive_keys = 1 (the per-step "
            "disturbance tables are derived from the raw controller key on the device)", params->disturb_kind);
    return covo_step_impl(h, params, args, key0, key1, f_disturb_shared, (hipStream_t)stream);
}

int covo_cholesky(covo_handle_t h, const float *A, int32_t n, int32_t batch, float *L_out, void *stream)
{
    REQUIRE(h, "covo_cholesky: null handle");
    REQUIRE(A && L_out && batch > 0, "covo_cholesky: bad argument");
    return launch_cholesky(A, n, batch, L_out, (hipStream_t)stream);
}

}  // extern "C"
