// hessian_adj.hip -- exact Hessian of the CoVO rollout objective by the second-order adjoint (gfx950, fp64).
//
// Replaces jax.jacfwd(jax.jacfwd(get_cumulated_cost)) of quadjax/controllers/covo.py:134-185:
//     C(a) = -( sum_{k<H} r(s_k) + r(s_0) ),  s_{k+1} = step_env(s_k, a_k, deterministic=True)
// (no discount, no done-freeze; reward on the PRE-step state; r(s_0) constant).  R = d^2 C / da^2.
//
// hessian.hip (kept as covo_hessian_pairs) runs one hyper-dual ROLLOUT per pair (i, j): exact, but its critical
// path is 31 hyper-dual steps (~1500 fp64 instructions each, one wave issues ~1 per 5 cycles) = 80 us.
// The same matrix follows from ONE hyper-dual step per (time, pair of step inputs) plus two short
// linear recursions.  With z_k = (x_k, u_k) in R^17 (state 13, raw action 4), x_{k+1} = f_k(z_k),
// J = sum_{k>=1} r_k(x_k):
//     costate    lam_31 = grad r_31,   lam_k = grad r_k + A_k^T lam_{k+1}          (A_k = df_k/dx)
//     sensitivity S_0 = 0,  S_{k+1} = A_k S_k + B_k E_k                          (S_k = dx_k/da, 13 x 128)
//     d^2 J/da^2 = sum_k [S_k; E_k]^T  Hess_z( r_k(x) + lam_{k+1} . f_k(z) )  [S_k; E_k]
// (E_k selects the four actions of step k).  All derivatives of the step come from evaluating the SAME
// templated model (quad_model.hpp) on hyper-dual numbers, so every JAX AD convention it mirrors (clip ties,
// |x|', the double clip of quadrotor.py:223/:258) carries over.  Three launches (KM rides in KC's):
//   KB  32 waves: primal rollout to step k (plain fp64), then the step on first-order duals (17 seeds) -> A_k, B_k, grad r_k
//   KC  9 waves: the sensitivity recursion (8 column tiles of S held in MFMA C/D layout, which IS the B operand
//       of the next step: four dependent v_mfma_f64_16x16x4_f64 per step, nothing leaves the registers) and
//       the costate recursion (the same product with A^T) -- no barriers, no sparsity assumptions
//   KM  32 x 153 lanes, 32 more workgroups of KC's launch: one hyper-dual step per pair (a <= b) of step inputs, contracted with
//       lam_{k+1} as soon as the costate chain of the same launch has stored it -> M_k = Hess_z(r_k + lam_{k+1}.f_k)
//   KD  36 lower 16x16 tiles: sum_k S_k^T Mxx_k S_k on the matrix cores (v_mfma_f64_16x16x4_f64; the product
//       Mxx S leaves the MFMA in exactly the register layout the next MFMA wants as its B operand), the
//       action blocks Mxu, Muu added in the epilogue.
// Critical path: 30 plain steps + 2 hyper-dual steps + a 31-step recursion of 13-vectors ~ 25 us.
// The kernels live in hessian_adj_body.hpp and are compiled twice: adj13 (x in R^13: the force of a step is a constant of that
// step -- none / gaussian / periodic / sin) and adj16 (x in R^16 = state + force: drag / mixed, whose next force depends on the
// PRE-step velocity and the current force; round 3 -- these two models used to take the 80 us per-pair kernel).
#include "covo_common.hpp"
#include "wave_reduce.hpp"
#include <cstdlib>
#include <cstring>
#include "sym_stats.hpp"
#include "disturb_model.hpp"
#include "step_begin.hpp"

typedef double f64x4 __attribute__((ext_vector_type(4)));

namespace {
struct AdjArgs {
    const float *state;     // [batch][COVO_STATE_FLOATS]
    const float *pos_traj;  // [T][3] shared
    const float *vel_traj;
    const float *a_mean;    // [batch][128]
    double *R;              // [batch][128][128]
    double *ws;             // [batch][WS_COUNT]
    int T;
    qm::Consts<double> c;            // shared model constants ...
    const qm::Consts<double> *cs;    // ... or one per batch entry (device, nullable): env instances with their own parameters
    size_t traj_stride;              // floats between the trajectories of consecutive batch entries (0: shared)
    SymStatsOut stats;               // rpart != null: KD also leaves the Sigma chain's input statistics of R there (every instance)
    const float4 *f_tab;             // [batch][H] per-step disturbance table (disturb.hip; wave-uniform kinds only) or null: no force after step 0
    int reward;                      // COVO_REWARD_*
    // adj16 only (drag / mixed: the force is part of the differentiated state); f_tab rows are then {g_k[3], c_k}
    double drag_k;                   // c_drag * (-|disturb_scale| / 1.5^2)   (free.py:41-47)
    double drag_off[3];              // disturb_params[:3] / 2
    const dm::Model *models;         // per batch entry next to cs (device, nullable): overrides drag_k / drag_off
    int *status;                     // the handle's sticky device status (nullable): COVO_DEVSTAT_ADJOINT when a costate wait times out
    // the step's begin work folded into KB (eager covo-online steps; launch_hessian's HessBegin): a_mean_raw != null -> KB reads the
    // UNSHIFTED mean through the shift's index map (covo.py:201-203) and its workgroup (0, 0) leaves what step_begin_kernel would
    // have left for the launches that follow: the shifted mean (a_mean points there: KC / KD read it a launch later), the per-step
    // scalars derived from the raw rng_act (step_begin.hpp) and the bumped sequence number
    const float *a_mean_raw;
    uint32_t *dyn_out;
    unsigned *seq;
    DynBlock blk;
    int derive_keys;
    float shared_noise_scale;
    int scan_prefix;                 // KB forms its primal prefix by parallel scans (adj13; COVO_HESS_SCAN=0: the sequential rollout)
};

__host__ __device__ inline double adj_drag_coeff(const dm::Model &m)
{
    return -(m.kind == COVO_DISTURB_DRAG ? 1.0 : (m.kind == COVO_DISTURB_MIXED ? 1.0 / 3.0 : 0.0)) * fabs((double)m.scale) / 2.25;
}

namespace adj13 {
#define ADJ_NX 13
#define ADJ_FS 0
#include "hessian_adj_body.hpp"
#undef ADJ_NX
#undef ADJ_FS
}  // namespace adj13
namespace adj16 {
#define ADJ_NX 16
#define ADJ_FS 1
#include "hessian_adj_body.hpp"
#undef ADJ_NX
#undef ADJ_FS
}  // namespace adj16
}  // namespace


constexpr size_t WS_COUNT_MAX = adj16::WS_COUNT > adj13::WS_COUNT ? adj16::WS_COUNT : adj13::WS_COUNT;
size_t hessian_workspace_bytes(int batch) { return (size_t)batch * WS_COUNT_MAX * sizeof(double); }

int launch_hessian(const float *state, const float *pos_traj, const float *vel_traj, int T, const covo_env_params &p,
                   const float *a_mean, int batch, double *R, void *workspace, hipStream_t s, const void *consts_dev,
                   size_t traj_stride, const SymStatsOut *stats, const float *f_tab, const void *models_dev, int *status_dev,
                   const HessBegin *begin)
{
    const bool fs = p.disturb_kind == COVO_DISTURB_DRAG || p.disturb_kind == COVO_DISTURB_MIXED;
    if (fs && (consts_dev != nullptr) != (models_dev != nullptr)) {
        covo_set_error("hessian: drag / mixed disturbance with per-instance constants needs the per-instance models too");
        return COVO_E_BADARG;
    }
    if (fs && f_tab == nullptr) {
        covo_set_error("hessian: disturb_kind=%d needs the per-step disturbance table (covo_disturb_table)", p.disturb_kind);
        return COVO_E_BADARG;
    }
    if ((p.disturb_kind == COVO_DISTURB_PERIODIC || p.disturb_kind == COVO_DISTURB_SIN) && f_tab == nullptr) {
        covo_set_error("hessian: disturb_kind=%d needs the per-step disturbance table (covo_disturb_table)", p.disturb_kind);
        return COVO_E_BADARG;
    }
    AdjArgs A;
    A.f_tab = (fs || p.disturb_kind == COVO_DISTURB_PERIODIC || p.disturb_kind == COVO_DISTURB_SIN) ? reinterpret_cast<const float4 *>(f_tab) : nullptr;
    const dm::Model m = dm::make_model(p);
    A.drag_k = adj_drag_coeff(m);
    for (int i = 0; i < 3; ++i) A.drag_off[i] = 0.5 * (double)m.dp[i];
    A.models = reinterpret_cast<const dm::Model *>(models_dev);
    A.reward = p.reward_kind;
    A.status = status_dev;
    A.state = state;
    A.pos_traj = pos_traj;
    A.vel_traj = vel_traj;
    A.a_mean = a_mean;
    A.a_mean_raw = nullptr;
    A.dyn_out = nullptr;
    A.seq = nullptr;
    std::memset(&A.blk, 0, sizeof(A.blk));
    A.derive_keys = 0;
    A.shared_noise_scale = 0.0f;
    if (begin != nullptr && batch == 1 && (g_dbg_hess_mask & 1)) {
        A.a_mean_raw = begin->a_mean_raw;
        A.dyn_out = begin->dyn_out;
        A.seq = begin->seq;
        std::memcpy(&A.blk, begin->blk, sizeof(A.blk));
        A.derive_keys = begin->derive_keys;
        A.shared_noise_scale = begin->shared_noise_scale;
    }
    static const int scan_prefix = [] { const char *v = std::getenv("COVO_HESS_SCAN"); return v ? std::atoi(v) : 1; }();
    A.scan_prefix = scan_prefix;
    A.R = R;
    A.ws = reinterpret_cast<double *>(workspace);
    A.T = T;
    A.c = make_consts<double>(p);
    A.cs = reinterpret_cast<const qm::Consts<double> *>(consts_dev);
    A.traj_stride = traj_stride;
    A.stats.rpart = nullptr;
    A.stats.fpart = nullptr;
    A.stats.diag = nullptr;
    A.stats.stride = 0;
    A.stats.flags = nullptr;
    A.stats.ready = nullptr;
    if (stats != nullptr) A.stats = *stats;
    // launch shapes: KB 32 waves; KC 9 chains + KM's 32 hyper-dual workgroups (which also contract with the costate); KD 36 tiles
    // batched: chains and hyper-dual workgroups as two launches (hessian_adj_body.hpp: adj_hd_kernel) -- 66.5 -> 62.5 us at 32 instances;
    // one instance keeps them in one launch (the hyper-dual workgroups start under the chains: 27.9 -> 25.7 us in round 3)
    const bool split_hd = batch >= 8;
#define ADJ_LAUNCH(NS, JAC)                                                                                                      \
    do {                                                                                                                          \
        if (g_dbg_hess_mask & 1) hipLaunchKernelGGL(JAC, dim3(NS::HH, batch), dim3(64), 0, s, A);                                 \
        if ((g_dbg_hess_mask & 2) && split_hd) {                                                                                  \
            hipLaunchKernelGGL(NS::adj_chain_kernel, dim3(9, batch), dim3(256), 0, s, A);                                         \
            hipLaunchKernelGGL(NS::adj_hd_kernel, dim3(NS::HH, batch), dim3(256), 0, s, A);                                       \
        } else if (g_dbg_hess_mask & 2) hipLaunchKernelGGL(NS::adj_chain_kernel, dim3(9 + NS::HH, batch), dim3(256), 0, s, A);    \
        if (g_dbg_hess_mask & 8) hipLaunchKernelGGL(NS::adj_gemm_kernel, dim3(36, batch), dim3(512), 0, s, A);                    \
    } while (0)
    if (fs) ADJ_LAUNCH(adj16, adj16::adj_jac_kernel<true>);
    else if (A.f_tab != nullptr) ADJ_LAUNCH(adj13, adj13::adj_jac_kernel<true>);
    else ADJ_LAUNCH(adj13, adj13::adj_jac_kernel<false>);
#undef ADJ_LAUNCH
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}

// host side of the per-instance constants: covo_env_params[n] -> qm::Consts<double>[n] (to be copied to the device)
size_t hessian_consts_bytes(int n) { return (size_t)n * sizeof(qm::Consts<double>); }
void hessian_fill_consts(const covo_env_params *params, int n, void *out)
{
    qm::Consts<double> *c = reinterpret_cast<qm::Consts<double> *>(out);
    for (int i = 0; i < n; ++i) c[i] = make_consts<double>(params[i]);
}
