// step_small.hip -- the WHOLE control step of the precomputed-Sigma controllers as ONE launch, for small sample counts (gfx950).
//
// SURVEY.md 8f-2 (GEMM -> clip -> rollout -> online softmax in one launch) where it pays: covo-offline and MPPI at N <= 16 384
// (BASELINE configs[0], [1]; the 8 192-sample shards of config [3]).  There the staged step is four launch boundaries
// (begin | noise | rollout + records | merge: 18 us at N = 1 024, 25 us at N = 8 192) around ~1 us of matrix work and a ~5 us
// rollout chain; at N = 65 536 the same fusion was built in round 2 and bought nothing (MFMA and VALU time add up on a SIMD,
// DESIGN.md 4.2) -- the gain here is the removed boundaries, not pipe overlap.  One workgroup = one 64-sample group = four waves:
//
//   phase 0  what step_begin_kernel does, per workgroup: shifted mean (covo.py:201-203) -> LDS; sampling key / MPPI's shared
//            disturbance from the raw rng_act (step_begin.hpp); the factor: covo-offline L_table[state.time] staged into the
//            padded LDS image (noise_gemm_body.hpp), MPPI the 32 4x4 block factors of the SHIFTED covariance (mppi.py:43-61)
//   phase 1  the group's actions a = clip(mu + L eps) (covo.py:212-224, mppi.py:53-66): covo-offline as the noise GEMM's two
//            32-sample tiles, each split over two waves (row tiles {0, 3} / {1, 2}) with in-register Philox -- the SAME device
//            functions and MFMA order as noise_gemm_kernel's SPLIT shape; MPPI per lane -> LDS [H][64] float4 (and the `a` work
//            buffer in HBM, write-only)
//   phase 2  the three-stage rollout pipeline (rollout_pipe.hpp: rp3_stages) on waves 0-2, stage A reading its stripes from
//            LDS; wave 3 shares the barriers
//   phase 3  the workgroup's online-softmax record (rollout_record, stripes from LDS) published with coherent stores; the
//            workgroup that takes the last ticket merges all records (softmax_merge.hpp: merge_body, the arithmetic of
//            merge_kernel) into the new mean -- or, on a sample-sharded rank, into the rank's record --, and shifts MPPI's
//            covariances in place (every workgroup has read them by then).
//
// Bit-identical to the staged step (same device functions, same record order, same merge): tests/test_gpu_parity.py
// test_small_fused_step_equals_staged.  Not taken (the staged launches run): position statistics, the per-step disturbance
// tables, the realworld reward, MPPI's covariance adaptation, more than 256 groups, the debug phase timers.
// Reference: quadjax/controllers/covo.py:201-278, mppi.py:43-129.
#include <cstring>
#include "rollout_common.hpp"
#include "noise_gemm_body.hpp"
#include "softmax_merge.hpp"
#include "step_begin.hpp"
#include "step_small.hpp"

constexpr int SS_BLOCK = 256;
constexpr int SS_CH = 2;

struct SmallLds {
    float4 a[COVO_H][COVO_WAVE];  // 32 KiB: the group's clipped actions, [t][sample]
    Rp3Lds<SS_CH> rings;          // 9 KiB
    float mus[COVO_NA];           // shifted mean
    uint32_t dyn[12];             // {key0, key1, f_shared[3], ...} as step_begin_kernel leaves them
    float rec_m[1], rec_s[1];
    __attribute__((aligned(16))) float rec_v[1][COVO_NA];
    int last;
    float mppi_L[COVO_H][16];     // MPPI: the block factors of the shifted covariance
};
// the padded image of L (covo-offline, phase 0/1) and the merge's scratch (phase 3) share the tail of the dynamic LDS
constexpr size_t SS_LDS_GEMM = sizeof(SmallLds) + (size_t)COVO_NA * NG_LDA * sizeof(float);
constexpr size_t SS_LDS_MPPI = sizeof(SmallLds) + sizeof(MergeLds);
static_assert(sizeof(MergeLds) <= (size_t)COVO_NA * NG_LDA * sizeof(float), "merge scratch must fit the factor image");

template <bool MPPI, bool DISC1, bool ROLL>
__global__ __launch_bounds__(SS_BLOCK) void step_small_kernel(const SmallStepArgs P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_raw[];
    SmallLds &S = *reinterpret_cast<SmallLds *>(ss_raw);
    float *Ls = reinterpret_cast<float *>(ss_raw + sizeof(SmallLds));       // [128][NG_LDA] (covo-offline)
    MergeLds &M = *reinterpret_cast<MergeLds *>(ss_raw + sizeof(SmallLds));  // phase 3
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    RolloutArgs A = P.R;
    const int N = A.N;

    // ---- phase 0
    const float *__restrict__ a_mean = P.a_mean_in;
    if (tid < COVO_NA) {
        if (P.dyn_mem != nullptr) {  // (a captured graph: the begin launch has shifted the mean of THIS replay; a_mean_in is not baked in)
            S.mus[tid] = P.a_mean_shift_out[tid];
        } else {
            const float v = (tid < COVO_NA - COVO_DU) ? a_mean[tid + COVO_DU] : a_mean[tid];  // covo.py:201-203
            S.mus[tid] = v;
            if (blockIdx.x == 0 && P.a_mean_shift_out != nullptr) P.a_mean_shift_out[tid] = v;
        }
    } else if (tid < COVO_NA + 4) {
        if (P.dyn_mem != nullptr) {  // a captured graph: the begin launch has left the step's scalars in device memory
            const int q = tid - COVO_NA;
            if (q == 0) { S.dyn[0] = P.dyn_mem[0]; S.dyn[1] = P.dyn_mem[1]; }
            else S.dyn[1 + q] = P.dyn_mem[1 + q];
        } else {
            step_begin_derive(tid - COVO_NA, P.blk, P.derive_keys, P.shared_noise_scale, S.dyn);
        }
    }
    if (MPPI) {
        if (tid >= SS_BLOCK - COVO_H) {  // the last 32 threads: block t of the SHIFTED covariance = old block t + 1 (mppi.py:43-49)
            const int t = tid - (SS_BLOCK - COVO_H);
            const float *src = P.mppi_cov + 16 * ((t < COVO_H - 1) ? t + 1 : t);
            float blk[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) blk[i] = src[i];
            mppi_factor_block(blk, &S.mppi_L[t][0]);
        }
    } else {
        const float *L = P.L_table;
        int t = __float_as_int(A.state[ST_TIME]);  // covo.py:107-108, clamped like a JAX gather
        t = t < 0 ? 0 : (t > P.n_table - 1 ? P.n_table - 1 : t);
        L += (size_t)t * COVO_NA * COVO_NA;
        ng_stage_factor<SS_BLOCK>(L, Ls, tid);
    }
    __syncthreads();
    const uint32_t k0 = S.dyn[0], k1 = S.dyn[1];
    // the rollouts' one shared disturbance vector (free.py:66-70 from the shared step key; 0 under deterministic = True)
    A.f_shared_dev = nullptr;
    A.f_shared[0] = __uint_as_float(S.dyn[2]);
    A.f_shared[1] = __uint_as_float(S.dyn[3]);
    A.f_shared[2] = __uint_as_float(S.dyn[4]);

    // which 64-sample group: the staged rollout's mapping (rollout_pipe3_kernel with one group per workgroup), so that record
    // i of this launch is record i of that one and the merge adds them in the same order
    int group = blockIdx.x;
    if (A.xcd_remap) {
        const int x = blockIdx.x & 7, m = (int)(blockIdx.x >> 3), q = A.xcd_remap;
        group = q * (x + 8 * (m / q)) + (m % q);
    }

    // ---- phase 1: the group's actions
    float4 *__restrict__ a_out = const_cast<float4 *>(A.a);
    if (MPPI) {
        // lane = sample, wave w takes steps 8 w .. 8 w + 7 (noise_blockdiag_kernel<PHILOX>'s arithmetic)
        const int n_raw = group * COVO_WAVE + lane;
        const int n = n_raw < N ? n_raw : N - 1;
#pragma unroll 2
        for (int i = 0; i < COVO_H / 4; ++i) {
            const int t = wave * (COVO_H / 4) + i;
            const float4 e = rngd::normal4((uint32_t)t, (uint64_t)(P.sample_offset + (int64_t)n), k0, k1);
            const float ev[4] = {e.x, e.y, e.z, e.w};
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) acc = fmaf((k <= r) ? S.mppi_L[t][r * 4 + k] : 0.0f, ev[k], acc);
                const float mt = S.mus[4 * t + r];
                o[r] = P.nanp ? qm::clip11_nan_(mt + acc) : qm::clip11_(mt + acc);  // mppi.py:66
            }
            const float4 v = make_float4(o[0], o[1], o[2], o[3]);
            S.a[t][lane] = v;
            if (n_raw < N) a_out[(size_t)t * N + n_raw] = v;
        }
    } else {
        const int j = lane & 31, kh = lane >> 5;
        const int tl = wave >> 1;  // which of the group's two 32-sample tiles
        const int n_raw = group * COVO_WAVE + tl * 32 + j;
        const int row = n_raw < N ? n_raw : N - 1;
        const uint64_t id = (uint64_t)(P.sample_offset + row);
        const float *__restrict__ La = Ls + j * NG_LDA + kh;
        f32x16 acc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rt][e] = 0.0f;
        auto store_rt = [&](int rt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int t = 8 * rt + 2 * g + kh;
                const float4 m4 = *reinterpret_cast<const float4 *>(S.mus + 4 * t);
                float4 v;
                if (P.nanp) {  // COVO_FLAG_PROPAGATE_NAN (covo.py:224 under jnp.clip)
                    v.x = qm::clip11_nan_(m4.x + acc[rt][4 * g + 0]);
                    v.y = qm::clip11_nan_(m4.y + acc[rt][4 * g + 1]);
                    v.z = qm::clip11_nan_(m4.z + acc[rt][4 * g + 2]);
                    v.w = qm::clip11_nan_(m4.w + acc[rt][4 * g + 3]);
                } else {
                    v.x = qm::clip11_(m4.x + acc[rt][4 * g + 0]);
                    v.y = qm::clip11_(m4.y + acc[rt][4 * g + 1]);
                    v.z = qm::clip11_(m4.z + acc[rt][4 * g + 2]);
                    v.w = qm::clip11_(m4.w + acc[rt][4 * g + 3]);
                }
                S.a[t][tl * 32 + j] = v;
                if (n_raw < N) a_out[(size_t)t * N + n_raw] = v;
            }
        };
        // noise_gemm_kernel's sequence for the row-tile set of this wave (work item 2 tile + half: odd -> {1, 2}, even -> {0, 3})
        auto seq = [&](auto mtag) {
            constexpr int MK = decltype(mtag)::value;
            BGroup b = gen_group(id, 0, kh, k0, k1);
            mfma_group<0, MK>(La, b, acc);
            if (MK & 1) store_rt(0);
            if (MK >> 1) b = gen_group(id, 1, kh, k0, k1);
            mfma_group<1, MK>(La, b, acc);
            if (MK & 2) store_rt(1);
            if (MK >> 2) b = gen_group(id, 2, kh, k0, k1);
            mfma_group<2, MK>(La, b, acc);
            if (MK & 4) store_rt(2);
            if (MK >> 3) b = gen_group(id, 3, kh, k0, k1);
            mfma_group<3, MK>(La, b, acc);
            if (MK & 8) store_rt(3);
        };
        if (wave & 1) seq(RtMask<6>());
        else seq(RtMask<9>());
    }
    __syncthreads();

    // ---- phase 2: the rollout (covo.py:227-263), stripes from LDS
    float cost = 0.0f;
    bool valid = false;
    int n = 0;
    float dummy_st[1][1][9];
    if (wave < 3)
        rp3_stages<DISC1, ROLL, SS_CH, -1, false, true, 0, 0, true>(A, S.rings, dummy_st, wave, 0, group, lane, &S.a[0][0], cost, valid, n);
    else
        rp3_idle_barriers<SS_CH>();

    // ---- phase 3: this group's record, then the last workgroup's merge (covo.py:266-278)
    const bool carrier = wave == 2;
    rollout_record<SS_BLOCK / COVO_WAVE, 1, true>(A, cost, valid && carrier, n, 0, carrier, lane, blockIdx.x, S.rec_m, S.rec_s,
                                                  S.rec_v, &S.a[0][0], true);
    // the write-through stores of the record are acknowledged before the ticket; the last workgroup merges (softmax_merge.hpp)
    A.merge_mean_old = S.mus;
    rollout_merge_last<SS_BLOCK>(A, M, S.last);
    if (!S.last) return;
    if (MPPI) {  // the in-place shift of a_cov (mppi.py:43-49): every workgroup took its factors from the old blocks long ago
        float blk[16];
        const int t = tid;
        if (t < COVO_H) {
            const float *src = P.mppi_cov + 16 * ((t < COVO_H - 1) ? t + 1 : t);
#pragma unroll
            for (int i = 0; i < 16; ++i) blk[i] = src[i];
        }
        __syncthreads();
        if (t < COVO_H) {
#pragma unroll
            for (int i = 0; i < 16; ++i) P.mppi_cov[16 * t + i] = blk[i];
        }
    }
}

bool step_small_eligible(const covo_ctx *h, const covo_env_params &p, const covo_step_args &a)
{
    if (a.mode != COVO_MODE_COVO_OFFLINE && a.mode != COVO_MODE_MPPI) return false;
    if (a.mode == COVO_MODE_MPPI && a.gamma_sigma != 0.0f) return false;      // second moments: reduce.hip's own stage 1
    if (a.pos_stats != nullptr) return false;                                  // covo.py:281's statistics: the STATS rollout
    if (p.reward_kind != COVO_REWARD_PENYAW) return false;
    if (p.disturb_kind != COVO_DISTURB_NONE && p.disturb_kind != COVO_DISTURB_GAUSSIAN) return false;  // per-step tables
    const int ng = (a.n_samples + COVO_WAVE - 1) / COVO_WAVE;
    return ng >= 1 && ng <= h->max_red_blocks && ng <= 256;
}

int launch_step_small(covo_ctx *h, const covo_env_params &p, const covo_step_args &a, const float *state, float *a_mean_shift,
                      const DynBlock *blk, const uint32_t *dyn_mem, float shared_noise_scale, unsigned *ticket, hipStream_t s)
{
    SmallStepArgs P;
    std::memset(&P, 0, sizeof(P));
    const int N = a.n_samples;
    fill_rollout_args(P.R, state, a.pos_traj, a.vel_traj, a.T, p, nullptr, a.a, N, h->cfg.discount, a.cost, nullptr, nullptr, nullptr,
                      nullptr, a.mode == COVO_MODE_MPPI ? 4 : 0);
    P.R.clip = 0;  // the stripes come straight from this launch's own clipped draw
    P.R.records = h->ws_partials;
    P.R.inv_lam = 1.0f / h->cfg.lam;
    P.R.merge_ticket = ticket;
    P.R.merge_final = a.partial_out == nullptr;
    P.R.merge_out = a.partial_out ? a.partial_out : a.a_mean;
    P.R.merge_gamma = a.gamma_mean;
    P.a_mean_in = a.a_mean_in ? a.a_mean_in : a.a_mean;
    P.a_mean_shift_out = a_mean_shift;
    P.L_table = a.L_table;
    P.n_table = a.n_table;
    P.mppi_cov = a.mode == COVO_MODE_MPPI ? a.a_cov : nullptr;
    P.sample_offset = a.sample_offset;
    if (blk != nullptr) P.blk = *blk;
    P.dyn_mem = dyn_mem;
    P.derive_keys = a.derive_keys;
    P.shared_noise_scale = shared_noise_scale;
    P.nanp = covo_propagate_nan(h) ? 1 : 0;
    const int ng = (N + COVO_WAVE - 1) / COVO_WAVE;
    static unsigned long long attr_devices = 0;  // (per device: covo_first_on_device)
    if (covo_first_on_device(attr_devices)) {
#define SS_ATTR(MPPI, D, R) COVO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(step_small_kernel<MPPI, D, R>), \
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MPPI ? SS_LDS_MPPI : SS_LDS_GEMM)))
        SS_ATTR(false, false, false); SS_ATTR(false, false, true); SS_ATTR(false, true, false); SS_ATTR(false, true, true);
        SS_ATTR(true, false, false); SS_ATTR(true, false, true); SS_ATTR(true, true, false); SS_ATTR(true, true, true);
#undef SS_ATTR
    }
    const bool mppi = a.mode == COVO_MODE_MPPI, disc1 = h->cfg.discount == 1.0f, roll = P.R.rollover != 0;
#define SS_GO(MPPI, D, R) hipLaunchKernelGGL((step_small_kernel<MPPI, D, R>), dim3(ng), dim3(SS_BLOCK), MPPI ? SS_LDS_MPPI : SS_LDS_GEMM, s, P)
    if (mppi) {
        if (disc1) { if (roll) SS_GO(true, true, true); else SS_GO(true, true, false); }
        else       { if (roll) SS_GO(true, false, true); else SS_GO(true, false, false); }
    } else {
        if (disc1) { if (roll) SS_GO(false, true, true); else SS_GO(false, true, false); }
        else       { if (roll) SS_GO(false, false, true); else SS_GO(false, false, false); }
    }
#undef SS_GO
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}
