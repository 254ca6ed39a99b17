// step.hip -- one C entry per MPC control step, replayed as a hipGraph.
//
// covo_mpc_step() enqueues everything quadjax's controller __call__ does between "shift the mean" and
// "new mean" (controllers/covo.py:201-275, mppi.py:43-125) for this rank's shard of samples on ONE
// stream, with no host work in between.  A control step is ~70 tiny launches for covo-online (the
// eigh-free Sigma pipeline alone is ~60); issued one by one from the host they leave ~150 us of gaps.
// The second call with the same buffers captures the sequence into a hipGraph, later calls replay it.
// Quantities that change every step (Philox key, MPPI's shared disturbance draw) live in a 32-byte
// device block refreshed by one async copy before each replay, so the captured kernel arguments stay valid.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "covo_common.hpp"
#include "eps_tiles.hpp"
#include "sym_stats.hpp"
#include "rng_device.hpp"
#include "step_begin.hpp"
#include "step_small.hpp"

// the ONE eager launch of every step, ahead of the replayed graph.  What changes per step travels in its kernel
// arguments (48 bytes; an async H2D copy of the same block runs as a ~5 us copy kernel on this stack): the controller's
// raw rng_act, the caller's shared disturbance, the address of the state.  It shifts the mean (covo.py:201-203), brings
// the state into the fixed-address buffer the captured launches read, and fills the device block the captured
// launches take their per-step scalars from -- with derive_keys, what the host would have computed from rng_act:
//   rng, act_key = split(rng_act); rng, step_key = split(rng)                (covo.py:212,225 / mppi.py:53,69)
//   MPPI: f_shared = scale * normal(split(split(split(step_key)[1])[0])[0], (3,))   (quadrotor.py:262, free.py:136,144)
__global__ void step_begin_kernel(const float *__restrict__ a_mean, float *__restrict__ a_mean_shift,
                                  uint32_t *__restrict__ dyn, float *__restrict__ state_buf, int derive_keys,
                                  float shared_noise_scale, const DynBlock blk, float *__restrict__ mppi_cov,
                                  float *__restrict__ mppi_Ls, unsigned *__restrict__ seq)
{
    if (threadIdx.x == 0 && seq != nullptr) seq[0] = seq[0] + 1u;  // the step's sequence number (the streamed finalize launch's flags)
    // MPPI (mppi_cov != null): the covariance shift + the 4x4 block factors ride in this launch (one launch less on a path
    // that is host bound at small N)
    if (mppi_cov != nullptr) mppi_prep(mppi_cov, mppi_Ls);
    const int i = threadIdx.x;  // 128 + 32 + 4 threads
    if (i < COVO_NA) {
        a_mean_shift[i] = (i < COVO_NA - COVO_DU) ? a_mean[i + COVO_DU] : a_mean[i];
    } else if (i < COVO_NA + COVO_STATE_FLOATS) {
        const float *src;
        __builtin_memcpy(&src, &blk.w[8], sizeof(src));
        state_buf[i - COVO_NA] = src[i - COVO_NA];
    } else {
        step_begin_derive(i - (COVO_NA + COVO_STATE_FLOATS), blk, derive_keys, shared_noise_scale, dyn);
    }
}

// covo_debug_time_step's copies of a step have no begin launch between them: the streamed finalize launch's flags would still
// carry the previous copy's sequence number (its workers would not wait for anything): one bump per copy
__global__ void stream_seq_bump_kernel(unsigned *seq) { seq[0] = seq[0] + 1u; }

struct StepKey {
    covo_step_args args;
    covo_env_params params;
    hipStream_t stream;
};

struct StepState {
    // device
    uint32_t *dyn;        // {key0, key1, f_shared[3] as float bits, pad[3], state pointer (8 bytes)}
    float *state_buf;     // [COVO_STATE_FLOATS] this step's state at a fixed address
    float *a_mean_shift;  // [128]
    double *R;            // [128][128]
    float *Sigma, *L;     // [128][128]
    float *Ls;            // [H][4][4] MPPI's block factors
    float4 *eps_tiled;    // covo-online: this step's epsilon in tile order, drawn under the Sigma chain (eps_tiles.hpp); or null
    float *f_tab_rollout, *f_tab_hess;  // [H][4] per-step disturbance tables of the sampling rollouts / the Hessian (disturb.hip)
    unsigned *ticket;     // arrival counter of the fused small step (step_small.hip); 0 between launches
    unsigned *sync;       // [16] the streamed finalize launch's sequence number and panel flags (StreamGemmArgs::sync)
    // graph cache
    bool have_key, have_graph;
    StepKey key;
    hipGraph_t graph;
    hipGraphExec_t exec;
};
constexpr int DYN_BYTES = 48;
// up to this many samples per GPU the step's epsilon is drawn by passenger workgroups of the Sigma chain's last launch
// (~6 us of work per 65 536 samples inside a ~30 us single-workgroup kernel); beyond, the GEMM draws it itself
constexpr int EPS_AHEAD_MAX_N = 262144;

static int step_state_init(covo_ctx *h)
{
    StepState *st = new StepState();
    std::memset(st, 0, sizeof(*st));
    COVO_CHECK_HIP(hipMalloc(&st->dyn, DYN_BYTES));
    COVO_CHECK_HIP(hipMalloc(&st->state_buf, COVO_STATE_FLOATS * sizeof(float)));
    COVO_CHECK_HIP(hipMalloc(&st->a_mean_shift, COVO_NA * sizeof(float)));
    COVO_CHECK_HIP(hipMalloc(&st->R, (size_t)COVO_NA * COVO_NA * sizeof(double)));
    COVO_CHECK_HIP(hipMalloc(&st->Sigma, (size_t)COVO_NA * COVO_NA * sizeof(float)));
    COVO_CHECK_HIP(hipMalloc(&st->L, (size_t)COVO_NA * COVO_NA * sizeof(float)));
    COVO_CHECK_HIP(hipMalloc(&st->Ls, COVO_H * 16 * sizeof(float)));
    COVO_CHECK_HIP(hipMalloc(&st->f_tab_rollout, COVO_H * 4 * sizeof(float)));
    COVO_CHECK_HIP(hipMalloc(&st->f_tab_hess, COVO_H * 4 * sizeof(float)));
    COVO_CHECK_HIP(hipMalloc(&st->ticket, sizeof(unsigned)));
    COVO_CHECK_HIP(hipMemset(st->ticket, 0, sizeof(unsigned)));
    COVO_CHECK_HIP(hipMalloc(&st->sync, 16 * sizeof(unsigned)));
    COVO_CHECK_HIP(hipMemset(st->sync, 0, 16 * sizeof(unsigned)));
    if (h->cfg.n_local <= EPS_AHEAD_MAX_N)
        COVO_CHECK_HIP(hipMalloc(&st->eps_tiled, (size_t)((h->cfg.n_local + 31) / 32) * 16 * 64 * sizeof(float4)));
    h->step = st;
    return 0;
}

void step_state_destroy(covo_ctx *h)
{
    StepState *st = reinterpret_cast<StepState *>(h->step);
    if (!st) return;
    if (st->have_graph) {
        (void)hipGraphExecDestroy(st->exec);
        (void)hipGraphDestroy(st->graph);
    }
    (void)hipFree(st->dyn);
    (void)hipFree(st->state_buf);
    (void)hipFree(st->a_mean_shift);
    (void)hipFree(st->R);
    (void)hipFree(st->Sigma);
    (void)hipFree(st->L);
    (void)hipFree(st->Ls);
    (void)hipFree(st->eps_tiled);
    (void)hipFree(st->f_tab_rollout);
    (void)hipFree(st->f_tab_hess);
    (void)hipFree(st->ticket);
    (void)hipFree(st->sync);
    delete st;
    h->step = nullptr;
}

int g_dbg_hess_mask = 15, g_dbg_sigma_stages = 4;
static const int g_dbg_eps_ahead = [] {  // COVO_EPS_AHEAD=0: the GEMM draws epsilon itself (A/B measurements)
    const char *v = std::getenv("COVO_EPS_AHEAD");
    return v ? std::atoi(v) : 1;
}();
// The experiment switches a new handle starts with (CovoOpts, covo_common.hpp), from the environment:
//   COVO_FUSE_SMALL=0     covo-offline and MPPI steps of <= 256 sample groups run their staged launches (begin | noise | rollout +
//                         records | merge) instead of the one fused launch of step_small.hip
//   COVO_STREAM_GEMM=0    covo-online's noise GEMM as a launch of its own behind the Sigma chain's finalize launch (rounds 1-4)
//                         instead of streamed under the factorisation inside it (sigma_ns.hip: ns_finalize_stream_kernel)
//   COVO_FOLD_BEGIN=0     eager covo-online steps keep the begin launch (default: its work rides in the Hessian's first launch)
//   COVO_NS_TAIL=sq,it    phases folded into the Sigma chain's persistent launches, as covo_debug_set_ns_tail
//   COVO_NS_MERGED=0      the Sigma chain's two persistent launches (squarings | iterations) as two launches (rounds 4-5) instead of one
//   COVO_NS_DEFLATE=0     the undeflated Newton-Schulz iteration;  COVO_NS_RITZ_INSIDE=0 / 2: the Ritz evaluations as one scan
//                         launch after the squarings / the filter's last iterate only (rounds 1-4's rule, timing reference)
// covo_debug_set_*(handle, ...) change them per handle afterwards (A/B measurements, parity tests).
CovoOpts covo_default_opts()
{
    auto env_int = [](const char *name, int dflt) {
        const char *v = std::getenv(name);
        return v ? std::atoi(v) : dflt;
    };
    CovoOpts o;
    o.fuse_small = env_int("COVO_FUSE_SMALL", 1);
    o.stream_gemm = env_int("COVO_STREAM_GEMM", 1);
    o.fold_begin = env_int("COVO_FOLD_BEGIN", 1);
    sigma_ns_tail_defaults(o);
    if (const char *v = std::getenv("COVO_NS_TAIL")) {  // "squarings,iterations" as covo_debug_set_ns_tail (scripts/tail_bench.py)
        int sq = -1, it = -1;
        if (std::sscanf(v, "%d,%d", &sq, &it) == 2 && sq >= 0 && it >= 0 && sq <= 64 && it <= 64) {
            o.ns_tail_squarings = o.ns_tail_squarings_batched = sq;
            o.ns_tail_iters = o.ns_tail_iters_batched = it;
        }
    }
    o.ns_deflate = env_int("COVO_NS_DEFLATE", 1) ? 1 : 0;
    o.ns_force_agent = 0;
    o.ns_merged = env_int("COVO_NS_MERGED", 1) ? 1 : 0;
    const int ri = env_int("COVO_NS_RITZ_INSIDE", 1);
    o.ns_ritz_inside = ri == 2 ? 2 : (ri ? 1 : 0);
    o.epoch = 0;
    return o;
}
static int g_dbg_step_mask = 63;  // (1: unused, the begin launch is not part of the graph) 2 Hessian, 4 Sigma, 8 noise GEMM, 16 rollout, 32 softmax update

// the launch sequence of one step (everything reads per-step scalars from st->dyn)
// begin != null (eager covo-online steps): no begin launch ran -- the Hessian's first launch does its work (HessBegin) and every
// launch reads the caller's state where it lies (state_direct) instead of the fixed-address copy
static int enqueue_step(covo_ctx *h, StepState *st, const covo_env_params &p, const covo_step_args &a, hipStream_t s,
                        const HessBegin *begin = nullptr, const float *state_direct = nullptr)
{
    const int M = g_dbg_step_mask;
    const int N = a.n_samples;
    const float *fdev = reinterpret_cast<const float *>(st->dyn + 2);
    float *am_shift = a.a_mean_shift ? a.a_mean_shift : st->a_mean_shift;
    int rc;
    const float *state = state_direct ? state_direct : st->state_buf;
    // covo-offline / MPPI at small N: noise -> rollout -> records -> merge as ONE launch (the begin launch has left the step's
    // scalars in st->dyn and the state in st->state_buf; MPPI: it has NOT touched a_cov, the fused launch shifts and factors)
    if (M == 63 && h->opt.fuse_small && step_small_eligible(h, p, a))
        return launch_step_small(h, p, a, state, am_shift, nullptr, st->dyn, 0.0f, st->ticket, s);
    // periodic / sin / drag / mixed (free.py:10-58): the wave-uniform part of every rollout step's force, for the sampling
    // rollouts (shared step key) and for the Hessian's deterministic rollout (per-step keys), resolved once per control step
    const bool tables = p.disturb_kind >= COVO_DISTURB_PERIODIC && p.disturb_kind <= COVO_DISTURB_MIXED;
    if (tables && (rc = launch_disturb_tables_step(p, state, st->dyn, a.rollout_deterministic, st->f_tab_rollout,
                                                   a.mode == COVO_MODE_COVO_ONLINE ? st->f_tab_hess : nullptr, s))) return rc;
    if (a.mode == COVO_MODE_COVO_ONLINE) {
        // the Hessian's last launch leaves the Sigma chain's input statistics in the chain's workspace: no prep launch
        const bool stats = (M & 2) && (M & 4) && (g_dbg_hess_mask & 15) == 15 && hessian_leaves_stats(p);
        const SymStatsOut so = sigma_ns_stats_out(h->ws_sigma);
        if ((M & 2) && (rc = launch_hessian(state, a.pos_traj, a.vel_traj, a.T, p, am_shift, 1, st->R, h->ws_hess, s, nullptr, 0,
                                            stats ? &so : nullptr, tables ? st->f_tab_hess : nullptr, nullptr, h->status_dev, begin)))
            return rc;  // :134-185
        float *Sig = a.a_cov ? a.a_cov : st->Sigma;
        // epsilon needs only the act key: it is drawn under the chain's single-workgroup finalize launch, the GEMM loads it
        const bool ahead = st->eps_tiled != nullptr && (M & 4) && g_dbg_sigma_stages >= 4 && g_dbg_eps_ahead;
        EpsGenArgs gen;
        gen.eps_tiled = ahead ? st->eps_tiled : nullptr;
        gen.dyn = st->dyn;
        gen.sample_offset = a.sample_offset;
        gen.N = N;
        gen.n_inst = 1;
        gen.dyn_stride = 0;
        gen.eps_stride = 0;
        // a_cov is written by the GEMM's first workgroups, not by the chain's one-workgroup finalize launch (CovDeferred)
        CovDeferred cov;
        std::memset(&cov, 0, sizeof(cov));
        const bool defer = (M & 8) && g_dbg_sigma_stages >= 4;
        // the GEMM streamed under the factorisation, inside the chain's last launch (one matrix, persistent launches allowed)
        StreamGemmArgs sg;
        sg.mu = am_shift;
        sg.dyn = st->dyn;
        sg.sample_offset = a.sample_offset;
        sg.N = N;
        sg.a_out = a.a;
        sg.L_stream = st->L;
        sg.sync = st->sync;
        sg.a_cov_out = Sig;
        sg.nanp = covo_propagate_nan(h) ? 1 : 0;
        const bool want_stream = h->opt.stream_gemm && (M & 4) && (M & 8) && g_dbg_sigma_stages >= 4;
        bool streamed = false;
        if ((M & 4) && (rc = launch_sigma_ns(h->opt, st->R, 1, a.sample_sigma, Sig, st->L, h->ws_sigma, s, &gen, h->status_dev,
                                             (h->cfg.flags & COVO_FLAG_SHARED_DEVICE) == 0, defer ? &cov : nullptr, stats,
                                             want_stream ? &sg : nullptr, &streamed))) return rc;
        if (streamed) {
        } else if (ahead) {
            if ((M & 8) && (rc = launch_noise_gemm(st->L, am_shift, reinterpret_cast<const float *>(st->eps_tiled), 0, 0,
                                                   a.sample_offset, N, a.a, s, nullptr, nullptr, 0, 1, true, &cov, covo_propagate_nan(h))))
                return rc;
        } else if ((M & 8) && (rc = launch_noise_gemm(st->L, am_shift, nullptr, 0, 0, a.sample_offset, N, a.a, s, st->dyn, nullptr, 0,
                                                      1, false, &cov, covo_propagate_nan(h))))
            return rc;
    } else if (a.mode == COVO_MODE_COVO_OFFLINE) {
        if ((M & 8) && (rc = launch_noise_gemm(a.L_table, am_shift, nullptr, 0, 0, a.sample_offset, N, a.a, s, st->dyn, state,
                                    a.n_table, 1, false, nullptr, covo_propagate_nan(h))))
            return rc;
    } else {  // MPPI: shift a_cov, factor the 4x4 blocks, per-step draws (mppi.py:43-66)
        // (a_cov was shifted and factored into st->Ls by the begin launch)
        if ((rc = launch_noise_blockdiag(st->Ls, am_shift, nullptr, 0, 0, a.sample_offset, N, a.a, s, st->dyn, covo_propagate_nan(h))))
            return rc;
    }
    const bool clipped = true;  // a comes straight from the noise kernels above
    // the rollout's workgroups leave the softmax update's stage-1 records themselves when they fit the merge (rollout.hip:
    // rollout_record); otherwise the stand-alone stage-1 kernel runs over the costs
    const int G = rollout_workgroups(N, a.pos_stats != nullptr);
    // MPPI's covariance adaptation needs second moments the in-rollout records do not carry: its own stage 1 (reduce.hip)
    const bool cov_adapt = a.mode == COVO_MODE_MPPI && a.gamma_sigma != 0.0f;
    const bool records = G <= h->max_red_blocks && !cov_adapt;
    if ((M & 16) && (rc = launch_rollout(state, a.pos_traj, a.vel_traj, a.T, p, nullptr, a.a, N, h->cfg.discount, clipped, a.cost,
                                         records ? nullptr : a.groupmin, a.pos_stats, h->ws_stats, s, fdev,
                                         records ? h->ws_partials : nullptr, h->cfg.lam, tables ? st->f_tab_rollout : nullptr,
                                         a.mode == COVO_MODE_MPPI ? 4 : 0, false)))  // MPPI's block-diagonal kernel: 256 samples per workgroup
        return rc;
    if (!(M & 32)) return 0;
    if (cov_adapt && a.partial_out != nullptr)  // a sample-sharded rank: its record with the second moments (836-float kind)
        return launch_softmax_reduce_cov(h, a.cost, a.a, N, a.groupmin, (N + 63) / 64, am_shift, a.partial_out, s);
    if (cov_adapt)  // mppi.py:109-125: new mean, then a_cov (already shifted by the begin launch) adapted in place
        return launch_softmax_update_cov(h, a.cost, a.a, N, a.groupmin, (N + 63) / 64, am_shift, a.gamma_mean, a.a_cov, a.gamma_sigma,
                                         a.a_mean, a.a_cov, s);
    if (records) {
        if (a.partial_out != nullptr) return launch_merge(h->ws_partials, G, h->cfg.lam, nullptr, 1.0f, nullptr, s, a.partial_out);
        return launch_merge(h->ws_partials, G, h->cfg.lam, am_shift, a.gamma_mean, a.a_mean, s);
    }
    // weights + update: finish locally, or leave this shard's record for the all-gather (covo.py:266-275)
    if (a.partial_out != nullptr)
        return launch_softmax_reduce(h, a.cost, a.a, N, a.groupmin, (N + 63) / 64, a.partial_out, nullptr, 1.0f, nullptr, s);
    return launch_softmax_reduce(h, a.cost, a.a, N, a.groupmin, (N + 63) / 64, nullptr, am_shift, a.gamma_mean,
                                 a.a_mean, s);
}

int covo_step_impl(covo_ctx *h, const covo_env_params *params, const covo_step_args *args, uint32_t key0, uint32_t key1,
                   const float *f_shared, hipStream_t s)
{
    if (!h->step) {
        int rc = step_state_init(h);
        if (rc) return rc;
    }
    if (h->dbg_epoch != h->opt.epoch) {  // a debug setter changed what a captured graph baked in (launch set, deflation switch)
        step_graphs_drop(h);
        h->dbg_epoch = h->opt.epoch;
    }
    StepState *st = reinterpret_cast<StepState *>(h->step);
    // per-step scalars: kernel arguments of the begin launch
    DynBlock blk;
    std::memset(&blk, 0, sizeof(blk));
    blk.w[0] = key0;
    blk.w[1] = key1;
    for (int i = 0; i < 3; ++i) {
        const float f = f_shared ? f_shared[i] : 0.0f;
        std::memcpy(&blk.w[2 + i], &f, 4);
    }
    std::memcpy(&blk.w[8], &args->state, sizeof(const float *));
    // the one shared gaussian vector of the sampling rollouts (free.py:66-70 from the shared step key): off under
    // step_env(deterministic=True) (quadrotor.py:234-235)
    const float shared_noise_scale =
        (params->disturb_kind == COVO_DISTURB_GAUSSIAN && !args->rollout_deterministic) ? params->dyn_noise_scale : 0.0f;
    // control_params.a_mean of this call: the handle's own buffer (a carried mean) or the caller's input (args->a_mean_in)
    const bool small = h->opt.fuse_small && step_small_eligible(h, *params, *args);
    if (small && (h->cfg.flags & COVO_FLAG_NO_GRAPH) != 0) {
        // an eager handle: the WHOLE step is one launch, the begin launch's work included (per workgroup, step_small.hip)
        st->have_key = false;  // (st->dyn / st->state_buf are not refreshed: a later graph capture starts from an eager call)
        return launch_step_small(h, *params, *args, args->state, args->a_mean_shift ? args->a_mean_shift : st->a_mean_shift, &blk,
                                 nullptr, shared_noise_scale, st->ticket, s);
    }
    // eager covo-online steps (no per-step force tables, whose launch precedes the Hessian and reads the scalars): the begin work
    // rides in the Hessian's first launch -- one launch boundary less (COVO_FOLD_BEGIN=0 keeps the begin launch)
    if (h->opt.fold_begin && args->mode == COVO_MODE_COVO_ONLINE && (h->cfg.flags & COVO_FLAG_NO_GRAPH) != 0 && g_dbg_hess_mask == 15 &&
        !(params->disturb_kind >= COVO_DISTURB_PERIODIC && params->disturb_kind <= COVO_DISTURB_MIXED)) {
        HessBegin hb;
        hb.a_mean_raw = args->a_mean_in ? args->a_mean_in : args->a_mean;
        hb.dyn_out = st->dyn;
        hb.seq = st->sync;
        hb.blk = &blk;
        hb.derive_keys = args->derive_keys;
        hb.shared_noise_scale = shared_noise_scale;
        st->have_key = false;
        return enqueue_step(h, st, *params, *args, s, &hb, args->state);
    }
    hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(COVO_NA + COVO_STATE_FLOATS + 4), 0, s,
                       args->a_mean_in ? args->a_mean_in : args->a_mean,
                       args->a_mean_shift ? args->a_mean_shift : st->a_mean_shift, st->dyn, st->state_buf, args->derive_keys,
                       shared_noise_scale, blk, (args->mode == COVO_MODE_MPPI && !small) ? args->a_cov : (float *)nullptr, st->Ls, st->sync);

    StepKey k;
    std::memset(&k, 0, sizeof(k));
    k.args = *args;
    k.args.state = nullptr;  // read through the dyn block: a new state address does not invalidate the graph
    k.args.a_mean_in = nullptr;  // read by the (eager) begin launch only
    k.params = *params;
    k.params.reset_traj = 0;  // the env step's auto-reset switches: no launch of the control step reads them
    k.params.reset_dt = k.params.reset_disturb_scale = 0.0;
    k.stream = s;
    const bool same = st->have_key && std::memcmp(&k, &st->key, sizeof(k)) == 0;
    if (same && st->have_graph) {
        COVO_CHECK_HIP(hipGraphLaunch(st->exec, s));
        return 0;
    }
    if (same && !st->have_graph && (h->cfg.flags & COVO_FLAG_NO_GRAPH) == 0) {
        // second call with identical buffers: capture (all one-time attribute calls / allocations happened in
        // the eager first call)
        // capture on the library's own stream (the caller's may be the legacy default stream, which cannot
        // capture); nothing executes during capture, the graph is then launched on the caller's stream
        hipStream_t cs = h->side_stream;
        COVO_CHECK_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue_step(h, st, *params, *args, cs);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(cs, &g);
        if (rc) return rc;
        if (e != hipSuccess) {
            covo_set_error("covo_mpc_step: stream capture failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        COVO_CHECK_HIP(hipGraphInstantiate(&st->exec, g, nullptr, nullptr, 0));
        st->graph = g;
        st->have_graph = true;
        COVO_CHECK_HIP(hipGraphLaunch(st->exec, s));
        return 0;
    }
    if (!same && st->have_graph) {  // buffers changed: drop the stale graph
        (void)hipGraphExecDestroy(st->exec);
        (void)hipGraphDestroy(st->graph);
        st->have_graph = false;
    }
    st->key = k;
    st->have_key = true;
    return enqueue_step(h, st, *params, *args, s);
}


// ---- profiling aid: `reps` copies of the selected part of one step captured into ONE graph and replayed; returns
// the average time per copy (GPU time between two events around the replay).  Inside a graph the launches cost
// what they cost in the product (no host launch overhead, no profiler inflation).  step_mask: enqueue_step's
// phases; hess_mask / sigma_stages: see covo_common.hpp.  The step must have been called once before (scratch).
int covo_debug_time_step_impl(covo_ctx *h, const covo_env_params *params, const covo_step_args *args, int step_mask,
                              int hess_mask, int sigma_stages, int reps, float *us_out, hipStream_t run)
{
    if (!h->step) {
        int rc = step_state_init(h);
        if (rc) return rc;
    }
    StepState *st = reinterpret_cast<StepState *>(h->step);
    hipStream_t cs = h->side_stream;
    const bool folded_online = h->opt.fold_begin && args->mode == COVO_MODE_COVO_ONLINE &&
                               !(params->disturb_kind >= COVO_DISTURB_PERIODIC && params->disturb_kind <= COVO_DISTURB_MIXED);
    if (((h->opt.fuse_small && step_small_eligible(h, *params, *args)) || folded_online) && (h->cfg.flags & COVO_FLAG_NO_GRAPH) != 0 &&
        args->state != nullptr) {
        // the last step ran without a begin launch (the one-launch small step; covo-online with the begin work folded into the
        // Hessian) and never filled the scratch the replayed launches read (state copy, shifted mean, keys; MPPI: shifted
        // covariance + block factors): one begin launch does, with the key the step would derive from (0, 0)
        DynBlock blk;
        std::memset(&blk, 0, sizeof(blk));
        std::memcpy(&blk.w[8], &args->state, sizeof(const float *));
        hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(COVO_NA + COVO_STATE_FLOATS + 4), 0, run,
                           args->a_mean_in ? args->a_mean_in : args->a_mean,
                           args->a_mean_shift ? args->a_mean_shift : st->a_mean_shift, st->dyn, st->state_buf, args->derive_keys, 0.0f,
                           blk, args->mode == COVO_MODE_MPPI ? args->a_cov : (float *)nullptr, st->Ls, st->sync);
        COVO_CHECK_HIP(hipStreamSynchronize(run));
    }
    g_dbg_step_mask = step_mask;
    g_dbg_hess_mask = hess_mask;
    g_dbg_sigma_stages = sigma_stages;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    int rc = 0;
    hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
        for (int r = 0; r < reps && !rc; ++r) {
            if (args->mode == COVO_MODE_COVO_ONLINE && h->opt.stream_gemm && (step_mask & 12) == 12)
                hipLaunchKernelGGL(stream_seq_bump_kernel, dim3(1), dim3(1), 0, cs, st->sync);
            rc = enqueue_step(h, st, *params, *args, cs);
        }
        e = hipStreamEndCapture(cs, &g);
    }
    g_dbg_step_mask = 63;
    g_dbg_hess_mask = 15;
    g_dbg_sigma_stages = 4;
    if (rc) return rc;
    COVO_CHECK_HIP(e);
    COVO_CHECK_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    COVO_CHECK_HIP(hipEventCreate(&e0));
    COVO_CHECK_HIP(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        COVO_CHECK_HIP(hipEventRecord(e0, run));
        COVO_CHECK_HIP(hipGraphLaunch(ge, run));
        COVO_CHECK_HIP(hipEventRecord(e1, run));
        COVO_CHECK_HIP(hipStreamSynchronize(run));
        float ms = 0.f;
        COVO_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    *us_out = best * 1e3f / (float)reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipGraphExecDestroy(ge);
    (void)hipGraphDestroy(g);
    return 0;
}


// =====================================================================================================================
// Env-batched covo-online step (BASELINE configs[4]: E independent env instances, each with its own state, reference
// trajectory, domain-randomised parameters, mean and noise key): ONE graph for all instances.  The latency-bound part
// -- Hessian and the eigh-free Sigma chain -- runs once for all E matrices (every kernel of both takes `batch`), so its
// ~50 launches are amortised over the instances; noise GEMM, rollout and softmax update are enqueued per instance.
// "Replicas only" (SURVEY.md 8e): no exchange between instances, env instances shard over GPUs without a collective.
struct BatchDyn {
    uint32_t w[COVO_MAX_ENVS][4];  // {rng_act[2] -> act_key[2]} per instance
};
__global__ void batch_set_dyn_kernel(uint32_t *__restrict__ dyn, const BatchDyn b, int n)
{
    const int i = threadIdx.x;
    if (i < 4 * n) dyn[12 * (i >> 2) + (i & 3)] = b.w[i >> 2][i & 3];
}
// per instance: shift the mean (covo.py:201-203); act_key = split(rng_act)[1] (covo.py:212); f_shared = 0 (deterministic)
__global__ void batch_begin_kernel(const float *__restrict__ a_mean, float *__restrict__ a_mean_shift, uint32_t *__restrict__ dyn)
{
    const int e = blockIdx.x, i = threadIdx.x;
    uint32_t *d = dyn + 12 * e;
    const uint32_t raw[2] = {d[0], d[1]};
    __syncthreads();
    if (i < COVO_NA) {
        a_mean_shift[e * COVO_NA + i] = (i < COVO_NA - COVO_DU) ? a_mean[e * COVO_NA + i + COVO_DU] : a_mean[e * COVO_NA + i];
    } else if (i == COVO_NA) {
        uint32_t k[2];
        host_split(raw, 1u, k);
        d[0] = k[0];
        d[1] = k[1];
        d[2] = d[3] = d[4] = 0u;
        d[10] = raw[0];  // the raw controller key, for the step's disturbance tables (disturb.hip)
        d[11] = raw[1];
    }
}

struct BatchState {
    int n_envs = 0;
    uint32_t *dyn = nullptr;        // [E][12]
    float *a_mean_shift = nullptr;  // [E][128]
    double *R = nullptr;            // [E][128][128]
    float *Sigma = nullptr, *L = nullptr;  // [E][128][128]
    void *consts = nullptr;         // qm::Consts<double>[E]   (Hessian)
    void *ro_args = nullptr;        // RolloutArgs[E]          (rollout)
    float *partials = nullptr;      // [E][max_red_blocks][COVO_PARTIAL_FLOATS]: the instances' softmax stage-1 records
    void *models = nullptr;         // dm::Model[E]            (disturbance tables, drag / mixed Hessian)
    float *tab_rollout = nullptr, *tab_hess = nullptr;  // [E][H][4] the step's disturbance tables (periodic / sin / drag / mixed)
    bool tables = false;            // the instances' disturbance model needs them
    void *env_inst = nullptr;       // EnvInst[env_inst_n] (env_step.hip): the per-instance constants of covo_env_step_batched
    int env_inst_n = 0;
    std::vector<covo_env_params> env_inst_params;
    std::vector<char> ro_args_host;
    std::vector<covo_env_params> params;
    covo_batch_args key;
    hipStream_t stream = nullptr;
    bool have_key = false, have_graph = false;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    float4 *eps_tiled = nullptr;  // [E][ceil(N/32)][16][64]: the step's epsilon of every instance, drawn under the Sigma chain's
    size_t eps_cap = 0;           // finalize launch (eps_tiles.hpp), as in the single step
};

static void batch_state_free(BatchState *b)
{
    if (b->have_graph) {
        (void)hipGraphExecDestroy(b->exec);
        (void)hipGraphDestroy(b->graph);
        b->have_graph = false;
    }
    (void)hipFree(b->dyn);
    (void)hipFree(b->a_mean_shift);
    (void)hipFree(b->R);
    (void)hipFree(b->Sigma);
    (void)hipFree(b->L);
    (void)hipFree(b->consts);
    (void)hipFree(b->ro_args);
    (void)hipFree(b->partials);
    (void)hipFree(b->models);
    (void)hipFree(b->tab_rollout);
    (void)hipFree(b->tab_hess);
    b->models = nullptr; b->tab_rollout = b->tab_hess = nullptr;
    b->dyn = nullptr; b->a_mean_shift = nullptr; b->R = nullptr; b->Sigma = b->L = nullptr; b->consts = nullptr;
    b->ro_args = nullptr; b->partials = nullptr;
}
// The captured graphs (fused step, env-batched step) hold the addresses of h->ws_sigma / h->ws_hess in their kernel nodes:
// whoever re-allocates a workspace (a larger batch through covo_sigma / covo_hessian / covo_mpc_step_batched) calls this
// first, so that a later covo_mpc_step re-captures instead of replaying launches that point into freed memory.
void step_graphs_drop(covo_ctx *h)
{
    StepState *st = reinterpret_cast<StepState *>(h->step);
    if (st && st->have_graph) {
        (void)hipGraphExecDestroy(st->exec);
        (void)hipGraphDestroy(st->graph);
        st->have_graph = false;
    }
    if (st) st->have_key = false;  // the next call runs eagerly, the one after captures again
    BatchState *b = reinterpret_cast<BatchState *>(h->batch);
    if (b && b->have_graph) {
        (void)hipGraphExecDestroy(b->exec);
        (void)hipGraphDestroy(b->graph);
        b->have_graph = false;
    }
    if (b) b->have_key = false;
}

// the device array of per-instance env constants for covo_env_step_batched, rebuilt only when the parameters change
int batch_env_inst(covo_ctx *h, const covo_env_params *params, int E, hipStream_t s, const void **inst_dev)
{
    BatchState *b = reinterpret_cast<BatchState *>(h->batch);
    if (!b) {
        b = new BatchState();
        h->batch = b;
    }
    const bool same = b->env_inst != nullptr && b->env_inst_n == E && (int)b->env_inst_params.size() == E &&
                      std::memcmp(b->env_inst_params.data(), params, (size_t)E * sizeof(covo_env_params)) == 0;
    if (!same) {
        COVO_CHECK_HIP(hipStreamSynchronize(s));  // launches that read the old array are done
        if (b->env_inst_n != E) {
            (void)hipFree(b->env_inst);
            b->env_inst = nullptr;
            b->env_inst_n = 0;
            COVO_CHECK_HIP(hipMalloc(&b->env_inst, env_step_inst_bytes(E)));
            b->env_inst_n = E;
        }
        std::vector<char> tmp(env_step_inst_bytes(E));
        env_step_fill_inst(params, E, tmp.data());
        COVO_CHECK_HIP(hipMemcpy(b->env_inst, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
        b->env_inst_params.assign(params, params + E);
    }
    *inst_dev = b->env_inst;
    return 0;
}

void batch_state_destroy(covo_ctx *h)
{
    BatchState *b = reinterpret_cast<BatchState *>(h->batch);
    if (!b) return;
    batch_state_free(b);
    (void)hipFree(b->eps_tiled);
    (void)hipFree(b->env_inst);  // (not in batch_state_free: the batched step re-allocates its scratch when the instance count
                                 // changes, possibly between batch_env_inst and the env step launch that reads this array)
    delete b;
    h->batch = nullptr;
}

// Seven batched launch sets for all E instances: begin, Hessian (4 kernels), Sigma chain (~47), noise GEMM, rollout,
// softmax partials, merge -- every kernel takes the instance as a grid dimension.  (Measured alternative, E = 32,
// N = 4096: per-instance GEMM/rollout/softmax launches 1 452 us per call; the same spread over 2 / 4 / 8 forked
// branches of the graph 1 103 / 1 140 / 1 153 us, while hipGraphLaunch's host cost grew from 47 to >300 us.  Round 4: the Sigma
// chain of the two halves of the instances on two forked branches, every phase its own launch: 72 700 control-steps/s against
// 71 200 unforked with the same launches and 76 300 with the persistent tails -- which must not run side by side: two persistent
// launches can starve each other of workgroup slots, the barriers then time out.)
static int batch_enqueue(covo_ctx *h, BatchState *b, const covo_batch_args &a, hipStream_t s)
{
    const int E = a.n_envs, N = a.n_samples;
    const int M = g_dbg_step_mask;  // 63 outside covo_debug_time_batched (which replays selected launch groups)
    int rc;
    if (M & 1) hipLaunchKernelGGL(batch_begin_kernel, dim3(E), dim3(COVO_NA + 64), 0, s, a.a_mean, b->a_mean_shift, b->dyn);
    // covo.py:231: CoVO's sampling rollouts run step_env(deterministic=True); get_hessian likewise (covo.py:152)
    if ((M & 1) && b->tables && (rc = launch_disturb_tables_batched(b->models, a.states, b->dyn, E, 1, b->tab_rollout, b->tab_hess, s)))
        return rc;
    // as in the single step: the Hessian's last launch leaves every instance's Sigma-chain input statistics, no prep launch
    const bool stats = (M & 2) && (M & 4) && (g_dbg_hess_mask & 15) == 15 && hessian_leaves_stats(b->params[0]);
    const SymStatsOut so = sigma_ns_stats_out(h->ws_sigma, E);
    if ((M & 2) && (rc = launch_hessian(a.states, a.pos_traj, a.vel_traj, a.T, b->params[0], b->a_mean_shift, E, b->R, h->ws_hess, s,
                                        b->consts, (size_t)a.T * 3, stats ? &so : nullptr, b->tables ? b->tab_hess : nullptr, b->models,
                                        h->status_dev)))
        return rc;
    float *Sig = a.a_cov ? a.a_cov : b->Sigma;
    // epsilon needs only the act keys: every instance's is drawn under the chain's finalize launch (32 of 256 CUs factor), the GEMM
    // loads it -- the in-kernel Philox costs the batched GEMM ~9 us, its matrix pipe hides no vector work
    const bool ahead = b->eps_tiled != nullptr && (M & 4) && (M & 8) && g_dbg_sigma_stages >= 4 && g_dbg_eps_ahead;
    EpsGenArgs gen;
    gen.eps_tiled = ahead ? b->eps_tiled : nullptr;
    gen.dyn = b->dyn;
    gen.sample_offset = 0;
    gen.N = N;
    gen.n_inst = E;
    gen.dyn_stride = 12;
    gen.eps_stride = (size_t)((N + 31) / 32) * 16 * 64;
    if ((M & 4) && (rc = launch_sigma_ns(h->opt, b->R, E, a.sample_sigma, Sig, b->L, h->ws_sigma, s, &gen, h->status_dev,
                                         (h->cfg.flags & COVO_FLAG_SHARED_DEVICE) == 0, nullptr, stats))) return rc;
    if (ahead) {
        if ((rc = launch_noise_gemm(b->L, b->a_mean_shift, reinterpret_cast<const float *>(b->eps_tiled), 0, 0, 0, N, a.a, s, nullptr,
                                    nullptr, 0, E, true, nullptr, covo_propagate_nan(h))))
            return rc;
    } else if ((M & 8) && (rc = launch_noise_gemm(b->L, b->a_mean_shift, nullptr, 0, 0, 0, N, a.a, s, b->dyn, nullptr, 0, E, false,
                                                  nullptr, covo_propagate_nan(h))))
        return rc;
    if ((M & 16) && (rc = launch_rollout_batched(b->ro_args_host.data(), b->ro_args, E, s))) return rc;
    if (!(M & 32)) return 0;
    const int G = rollout_workgroups(N, false, E);
    if (G <= h->max_red_blocks)  // the rollout's workgroups have left the records (rollout_record): instance e's are [e][G]
        return launch_merge(b->partials, G, h->cfg.lam, b->a_mean_shift, a.gamma_mean, a.a_mean, s, nullptr, E);
    return launch_softmax_reduce(h, a.cost, a.a, N, a.groupmin, (N + 63) / 64, nullptr, b->a_mean_shift, a.gamma_mean, a.a_mean, s,
                                 b->partials, E);
}

// profiling aid (bench.py --config envs): `reps` copies of the selected launch groups of the LAST covo_mpc_step_batched call in
// one graph; GPU microseconds per copy.  step_mask as in covo_debug_time_step (1 begin, 2 Hessian, 4 Sigma, 8 GEMM, 16 rollout,
// 32 update).  The copies re-read the same means and states (the begin launch is normally left out: it would re-split the keys).
int covo_debug_time_batched_impl(covo_ctx *h, int step_mask, int reps, float *us_out, hipStream_t run)
{
    BatchState *b = reinterpret_cast<BatchState *>(h->batch);
    if (!b || !b->have_key) {
        covo_set_error("covo_debug_time_batched: call covo_mpc_step_batched first");
        return COVO_E_BADARG;
    }
    hipStream_t cs = h->side_stream;
    COVO_CHECK_HIP(hipStreamSynchronize(run));
    g_dbg_step_mask = step_mask;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    int rc = 0;
    hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
        for (int r = 0; r < reps && !rc; ++r) rc = batch_enqueue(h, b, b->key, cs);
        e = hipStreamEndCapture(cs, &g);
    }
    g_dbg_step_mask = 63;
    if (rc) return rc;
    COVO_CHECK_HIP(e);
    COVO_CHECK_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    COVO_CHECK_HIP(hipEventCreate(&e0));
    COVO_CHECK_HIP(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        COVO_CHECK_HIP(hipEventRecord(e0, run));
        COVO_CHECK_HIP(hipGraphLaunch(ge, run));
        COVO_CHECK_HIP(hipEventRecord(e1, run));
        COVO_CHECK_HIP(hipStreamSynchronize(run));
        float ms = 0.f;
        COVO_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    *us_out = best * 1e3f / (float)reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipGraphExecDestroy(ge);
    (void)hipGraphDestroy(g);
    return 0;
}

int covo_step_batched_impl(covo_ctx *h, const covo_batch_args *args, const covo_env_params *params, const uint32_t *keys,
                           hipStream_t s)
{
    const int E = args->n_envs;
    if (h->dbg_epoch != h->opt.epoch) {  // as in covo_step_impl
        step_graphs_drop(h);
        h->dbg_epoch = h->opt.epoch;
    }
    BatchState *b = reinterpret_cast<BatchState *>(h->batch);
    if (!b) {
        b = new BatchState();
        h->batch = b;
    }
    bool same = b->have_key && b->n_envs == E && std::memcmp(&b->key, args, sizeof(*args)) == 0 && b->stream == s &&
                std::memcmp(b->params.data(), params, (size_t)E * sizeof(covo_env_params)) == 0;
    if (!same) {
        // new buffers / parameters / instance count: (re)allocate scratch and drop the stale graph (outside the steady state)
        COVO_CHECK_HIP(hipStreamSynchronize(s));
        if (b->have_graph) {
            (void)hipGraphExecDestroy(b->exec);
            (void)hipGraphDestroy(b->graph);
            b->have_graph = false;
        }
        if (b->n_envs != E) {
            batch_state_free(b);
            b->have_graph = false;
            const size_t M = (size_t)COVO_NA * COVO_NA;
            COVO_CHECK_HIP(hipMalloc(&b->dyn, (size_t)E * 12 * sizeof(uint32_t)));
            COVO_CHECK_HIP(hipMalloc(&b->a_mean_shift, (size_t)E * COVO_NA * sizeof(float)));
            COVO_CHECK_HIP(hipMalloc(&b->R, (size_t)E * M * sizeof(double)));
            COVO_CHECK_HIP(hipMalloc(&b->Sigma, (size_t)E * M * sizeof(float)));
            COVO_CHECK_HIP(hipMalloc(&b->L, (size_t)E * M * sizeof(float)));
            COVO_CHECK_HIP(hipMalloc(&b->consts, hessian_consts_bytes(E)));
            COVO_CHECK_HIP(hipMalloc(&b->ro_args, rollout_args_bytes(E)));
            COVO_CHECK_HIP(hipMalloc(&b->partials, (size_t)E * h->max_red_blocks * COVO_PARTIAL_FLOATS * sizeof(float)));
            COVO_CHECK_HIP(hipMalloc(&b->models, disturb_models_bytes(E)));
            COVO_CHECK_HIP(hipMalloc(&b->tab_rollout, (size_t)E * COVO_H * 4 * sizeof(float)));
            COVO_CHECK_HIP(hipMalloc(&b->tab_hess, (size_t)E * COVO_H * 4 * sizeof(float)));
            b->n_envs = E;
        }
        b->params.assign(params, params + E);
        std::vector<char> tmp(hessian_consts_bytes(E));
        hessian_fill_consts(params, E, tmp.data());
        COVO_CHECK_HIP(hipMemcpy(b->consts, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
        tmp.assign(disturb_models_bytes(E), 0);
        disturb_fill_models(params, E, tmp.data());
        COVO_CHECK_HIP(hipMemcpy(b->models, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
        b->tables = params[0].disturb_kind >= COVO_DISTURB_PERIODIC && params[0].disturb_kind <= COVO_DISTURB_MIXED;
        b->ro_args_host.assign(rollout_args_bytes(E), 0);
        const int N = args->n_samples, ng = (N + 63) / 64;
        const int bG = rollout_workgroups(N, false, E);
        const bool brec = bG <= h->max_red_blocks;
        for (int e = 0; e < E; ++e)
            rollout_fill_args(b->ro_args_host.data(), e, args->states + (size_t)e * COVO_STATE_FLOATS,
                              args->pos_traj + (size_t)e * args->T * 3, args->vel_traj + (size_t)e * args->T * 3, args->T,
                              params[e], args->a + (size_t)e * COVO_H * N * 4, N, h->cfg.discount, args->cost + (size_t)e * N,
                              brec ? nullptr : args->groupmin + (size_t)e * ng, reinterpret_cast<const float *>(b->dyn + 12 * e + 2),
                              brec ? b->partials + (size_t)e * bG * COVO_PARTIAL_FLOATS : nullptr, h->cfg.lam, true,
                              b->tables ? b->tab_rollout + (size_t)e * COVO_H * 4 : nullptr);
        COVO_CHECK_HIP(hipMemcpy(b->ro_args, b->ro_args_host.data(), b->ro_args_host.size(), hipMemcpyHostToDevice));
        {
            const size_t need_e = (size_t)E * ((N + 31) / 32) * 16 * 64;
            if (need_e > b->eps_cap) {
                (void)hipFree(b->eps_tiled);
                b->eps_tiled = nullptr;
                b->eps_cap = 0;
                COVO_CHECK_HIP(hipMalloc(&b->eps_tiled, need_e * sizeof(float4)));
                b->eps_cap = need_e;
            }
        }
        const size_t need_s = sigma_ns_workspace_bytes(E), need_h = hessian_workspace_bytes(E);
        if (need_s > h->ws_sigma_bytes || need_h > h->ws_hess_bytes) step_graphs_drop(h);  // captured launches point into them
        if (need_s > h->ws_sigma_bytes) {
            (void)hipFree(h->ws_sigma);
            h->ws_sigma = nullptr;
            h->ws_sigma_bytes = 0;
            COVO_CHECK_HIP(hipMalloc(&h->ws_sigma, need_s));
            h->ws_sigma_bytes = need_s;
        }
        if (need_h > h->ws_hess_bytes) {
            (void)hipFree(h->ws_hess);
            h->ws_hess = nullptr;
            h->ws_hess_bytes = 0;
            COVO_CHECK_HIP(hipMalloc(&h->ws_hess, need_h));
            h->ws_hess_bytes = need_h;
        }
        b->key = *args;
        b->stream = s;
        b->have_key = true;
    }
    BatchDyn blk;
    std::memset(&blk, 0, sizeof(blk));
    for (int e = 0; e < E; ++e) {
        blk.w[e][0] = keys[2 * e];
        blk.w[e][1] = keys[2 * e + 1];
    }
    hipLaunchKernelGGL(batch_set_dyn_kernel, dim3(1), dim3(256), 0, s, b->dyn, blk, E);
    if (b->have_graph) {
        COVO_CHECK_HIP(hipGraphLaunch(b->exec, s));
        return 0;
    }
    if (same && (h->cfg.flags & COVO_FLAG_NO_GRAPH) == 0) {  // second identical call: capture
        hipStream_t cs = h->side_stream;
        COVO_CHECK_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        const int rc = batch_enqueue(h, b, *args, cs);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(cs, &g);
        if (rc) return rc;
        if (e != hipSuccess) {
            covo_set_error("covo_mpc_step_batched: stream capture failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        COVO_CHECK_HIP(hipGraphInstantiate(&b->exec, g, nullptr, nullptr, 0));
        b->graph = g;
        b->have_graph = true;
        COVO_CHECK_HIP(hipGraphLaunch(b->exec, s));
        return 0;
    }
    return batch_enqueue(h, b, *args, s);
}

// test hook: the Hessians of the LAST batched step (E x 128 x 128 doubles), device -> host
int covo_debug_batched_hessians_impl(covo_ctx *h, double *out, int64_t offset_doubles, int64_t count, hipStream_t s)
{
    BatchState *b = reinterpret_cast<BatchState *>(h->batch);
    if (b == nullptr || b->R == nullptr) {
        covo_set_error("covo_debug_batched_hessians: no batched step has run on this handle");
        return COVO_E_BADARG;
    }
    COVO_CHECK_HIP(hipMemcpyAsync(out, b->R + offset_doubles, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, s));
    return 0;
}
