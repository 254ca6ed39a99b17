// rollout_common.hpp -- argument block, softmax-record epilogue and the pipelined kernel shared by the translation units that
// instantiate it (rollout.hip: the default reward / disturbance family; rollout_var_r0.hip, rollout_var_r1.hip: the other
// reward and disturbance variants -- split only to compile them in parallel).
#pragma once
#include "covo_common.hpp"
#include "softmax_merge.hpp"

struct RolloutArgs {
    const float *state;
    const float *pos_traj;
    const float *vel_traj;
    const float4 *a;   // [H][N]
    float *cost;       // [N]
    float *groupmin;   // [ceil(N/64)] per-wave cost minima, or null
    double *stats_ws;  // [gridDim.x][H*6] or null
    int N, T, max_steps;
    float discount;
    float f_shared[3];
    const float *f_shared_dev;  // nullable: {fx, fy, fz} in device memory (graph replays), overrides f_shared
    int xcd_remap;              // > 0: workgroup -> sample chunks follow the noise GEMM's XCD placement; = 64-sample groups per GEMM workgroup
    float *records;             // nullable: [workgroups][COVO_PARTIAL_FLOATS] online-softmax records (rollout_record below)
    float inv_lam;
    int clip;      // 1: re-apply step_env's clip to the stripes (quadrotor.py:223,258); 2: the same with jnp.clip's NaN propagation
                   // (COVO_FLAG_PROPAGATE_NAN); 0: the producer guarantees clipped stripes
    int rollover;  // 1: is_terminal's rollover test is on (quadrotor.py:486-490)
    int reward;    // COVO_REWARD_*: selects the REWARD template variant
    int fdist;     // FDIST template variant (rollout_pipe.hpp): 0 one vector for all steps >= 1, 1 per-step table, 2 per-sample force
    const float4 *f_tab;  // fdist 1 / 2: [H] rows {g_k[3], c_k} (disturb.hip)
    float drag_k;         // fdist 2: c_drag * (-|disturb_scale| / 1.5^2)  (free.py:41-47)
    float drag_off[3];    // fdist 2: disturb_params[:3] / 2
    qm::Consts<float> c;
    // the one-launch small step only (step_small.hip; null elsewhere): the workgroup that takes the last ticket also merges all records (softmax_merge.hpp:
    // merge_kernel's arithmetic) -- the softmax update finishes inside this launch, no merge launch follows.  The records are then
    // published with coherent stores.  merge_final: merge_out[128] = the new mean blended with merge_mean_old (covo.py:270-275);
    // else merge_out[130] = the merged record (a sample-sharded rank)
    unsigned *merge_ticket;
    float *merge_out;
    const float *merge_mean_old;
    float merge_gamma;
    int merge_final;
};

// rollout.hip: the argument block of one rollout over N samples (the producer of the stripes is the noise GEMM unless
// xcd_groups says otherwise); records / clip are set by the caller
void fill_rollout_args(RolloutArgs &A, const float *state, const float *pos_traj, const float *vel_traj, int T,
                       const covo_env_params &p, const float *f_shared, const float *a, int N, float discount, float *cost,
                       float *groupmin, double *stats_ws, const float *f_shared_dev, const float *f_tab, int xcd_groups = 0,
                       int nbatch = 1);


// scripts/probe/rollout_probe.hip compiles this file with ROLLOUT_PROBE: every workgroup leaves {XCC, HW_ID, start, end}
// (s_memrealtime, 100 MHz) -- where the dispatcher put it and when it ran.  Compiled out of the library.
#ifdef ROLLOUT_PROBE
__device__ unsigned long long *g_ro_probe;
#define RO_PROBE_BEGIN()                                                                                   \
    unsigned long long ro_t0_ = wall_clock64();
#define RO_PROBE_END(TID)                                                                                  \
    if (threadIdx.x == (TID) && g_ro_probe) {                                                                  \
        unsigned xcc_, hw_;                                                                                \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                  \
        unsigned long long *o_ = g_ro_probe + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);           \
        o_[0] = xcc_; o_[1] = hw_; o_[2] = ro_t0_; o_[3] = wall_clock64();                                 \
    }
#else
#define RO_PROBE_BEGIN()
#define RO_PROBE_END(TID)
#endif

__device__ __forceinline__ float lane_bcast(float v, int lane)  // v_readlane_b32 -> SGPR operand
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// The softmax update's first stage, done by the workgroup that has just produced the costs (fused step): one
// online-softmax record {m, s, v[128]} per workgroup -- m = its cost minimum, s = sum_n w_n, v = sum_n w_n a_n with
// w_n = exp(-(c_n - m)/lam) (covo.py:266-272 with the LOCAL minimum; merge_kernel rescales every record by
// exp(-(m - min_g m_g)/lam), the same merge that combines the per-GPU records of a sample-sharded step, so the result
// is the reference's softmax to fp32 rounding).  Saves the launch and the cost re-read of softmax_partial_kernel; the
// stripes of the few samples with a non-zero weight (at lam = 0.01 a weight underflows once c - m > 1.04) are re-read
// from the L2 that has just served them.  Called by every wave of the workgroup: NWAVES waves in all, NW of them carry
// costs (`carrier`, slot `wave` < NW); the others pass valid = false and only take part in the barriers and the final sums.
// A_LDS (the fused small step): the workgroup's ONE 64-sample group has its stripes in LDS (a_lds [H][64] float4; lane l owns
// sample l); coh (wave-uniform): the record is read by another workgroup of the SAME launch (the last arriver merges): agent-scope
// relaxed atomic stores (write-through, coherent across the XCDs' L2s), as the Sigma chain's persistent launches publish their tiles.
template <int NWAVES, int NW, bool A_LDS = false>
__device__ __forceinline__ void rollout_record(const RolloutArgs &A, float cost, bool valid, int n, int wave, bool carrier, int lane,
                                               int wg, float *s_m, float *s_s, float (*s_v)[COVO_NA],
                                               const float4 *__restrict__ a_lds = nullptr, const bool coh = false)
{
    // (round 4: the waves that carry no cost -- two of three per SIMD in the pipelined kernel -- only take part in the two
    // barriers and the final stores; they used to run the whole epilogue on zero weights next to the one wave that matters)
    if (carrier) {
        const float wm = wave_min(valid ? cost : __builtin_inff());
        if (lane == 0) s_m[wave] = wm;
    }
    __syncthreads();
    float m = s_m[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) m = fminf(m, s_m[i]);
    if (carrier) {
    const float w = valid ? expf((m - cost) * A.inv_lam) : 0.0f;
    const float sw = wave_sum(w);
    unsigned long long live = __ballot(w > 0.0f);
    const int t = lane & 31, half = lane >> 5;  // lane -> action stripe t; the two half-waves take alternate live samples
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    while (live != 0ull) {
        // up to 8 live samples per trip (4 per half-wave): their stripe loads are all issued before the first is used --
        // one L2 round trip per trip, not per sample
        float wv[4];
        float4 av[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int l0 = 0, l1 = 0;
            bool one = false, two = false;
            if (live != 0ull) {
                l0 = (int)__builtin_ctzll(live);
                live &= live - 1ull;
                one = true;
            }
            if (live != 0ull) {
                l1 = (int)__builtin_ctzll(live);
                live &= live - 1ull;
                two = true;
            }
            const int l = half ? l1 : l0;
            wv[q] = __shfl(w, l, COVO_WAVE);
            const int nl = __shfl(n, l, COVO_WAVE);
            if (half ? !two : !one) wv[q] = 0.0f;
            av[q] = A_LDS ? a_lds[t * COVO_WAVE + l] : A.a[(size_t)t * A.N + nl];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc.x = fmaf(wv[q], av[q].x, acc.x);
            acc.y = fmaf(wv[q], av[q].y, acc.y);
            acc.z = fmaf(wv[q], av[q].z, acc.z);
            acc.w = fmaf(wv[q], av[q].w, acc.w);
        }
    }
    acc.x += __shfl_xor(acc.x, 32, COVO_WAVE);
    acc.y += __shfl_xor(acc.y, 32, COVO_WAVE);
    acc.z += __shfl_xor(acc.z, 32, COVO_WAVE);
    acc.w += __shfl_xor(acc.w, 32, COVO_WAVE);
    if (half == 0) *reinterpret_cast<float4 *>(&s_v[wave][4 * t]) = acc;
    if (lane == 0) s_s[wave] = sw;
    }
    __syncthreads();
    float *rec = A.records + (size_t)wg * COVO_PARTIAL_FLOATS;
    const int tid = threadIdx.x;
    if (tid < COVO_NA) {
        float v = s_v[0][tid];
#pragma unroll
        for (int i = 1; i < NW; ++i) v += s_v[i][tid];
        if (coh) __hip_atomic_store(rec + 2 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else rec[2 + tid] = v;
    }
    if (tid == 0) {
        float ss = s_s[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) ss += s_s[i];
        if (coh) {
            __hip_atomic_store(rec + 0, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(rec + 1, ss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            rec[0] = m;
            rec[1] = ss;
        }
    }
}

// (step_small.hip only since round 6; the same inside the product rollout launch measured slower at every size and was removed:
// scripts/probe/rollout_merge_in_launch.hpp)  The update's second stage inside the launch that left the records: every workgroup takes a ticket once its (coherently stored)
// record is acknowledged; the one that takes the last merges all gridDim.x records.  atomicInc wraps to 0 at the last arrival:
// the counter re-arms itself for the next launch.  Called by every thread of every workgroup (barriers); THREADS = blockDim.x.
template <int THREADS>
__device__ __forceinline__ void rollout_merge_last(const RolloutArgs &A, MergeLds &M, int &last_flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last_flag = (atomicInc(A.merge_ticket, gridDim.x - 1) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!last_flag) return;
    if (A.merge_final) merge_body<THREADS, true, true>(A.records, (int)gridDim.x, A.inv_lam, A.merge_mean_old, A.merge_gamma, A.merge_out, COVO_PARTIAL_FLOATS, M);
    else merge_body<THREADS, false, true>(A.records, (int)gridDim.x, A.inv_lam, nullptr, 1.0f, A.merge_out, COVO_PARTIAL_FLOATS, M);
}

#ifdef ROLLOUT_LAB_BASELINE  // the one-lane-per-sample kernel of round 1: only scripts/probe/rollout_lab.hip still compiles it
constexpr int RO_BLOCK = 256;
// STATS: accumulate per-step position sums (covo.py:281).  DISC1: discount == 1 (skip the multiply).
// CLIP: re-apply step_env's clip (quadrotor.py:223,258); off when the producer guarantees clipped
// stripes.  PF: how many action stripes are in flight (32 = the whole horizon is issued up front:
// right at <= 2 waves/SIMD where nothing else hides HBM latency; 8 keeps VGPRs low for big N).
// BATCHED (env-batched step): workgroup row blockIdx.y rolls out instance y, whose argument block -- own state,
// trajectory, parameters, action stripes, cost slice -- is batch[y] in device memory (wave-uniform scalar loads).
template <bool STATS, bool DISC1, bool CLIP, int PF, bool BATCHED = false>
__global__ __launch_bounds__(RO_BLOCK) void rollout_kernel(const RolloutArgs A_, const RolloutArgs *__restrict__ batch)
{
    const RolloutArgs &A = BATCHED ? batch[blockIdx.y] : A_;
    RO_PROBE_BEGIN();
    __shared__ float rec_m[RO_BLOCK / COVO_WAVE], rec_s[RO_BLOCK / COVO_WAVE];
    __shared__ __attribute__((aligned(16))) float rec_v[RO_BLOCK / COVO_WAVE][COVO_NA];
    __shared__ double sacc[STATS ? (RO_BLOCK / COVO_WAVE) * COVO_H * 6 : 1];  // one slot per wave: no atomics, fixed order
    __shared__ float spanel[STATS ? (RO_BLOCK / COVO_WAVE) * 8 * 3 * COVO_WAVE : 1];  // 8 steps x 3 axes x 64 lanes per wave
    const int tid = threadIdx.x, lane = tid & (COVO_WAVE - 1), wave = tid / COVO_WAVE;
    const float *__restrict__ st = A.state;
    const int time0 = __float_as_int(st[ST_TIME]);

    // XCD affinity (speed only, never correctness): the noise GEMM's workgroup g writes the stripes of samples
    // [128 g, 128 g + 128) and runs on XCD g % 8 (observed dispatch order); this workgroup (XCD blockIdx % 8) takes
    // the two 128-sample chunks g = x + 16 i and x + 16 i + 8 (x = blockIdx % 8, i = blockIdx / 8), so its reads hit
    // the 4 MiB L2 that has just absorbed those writes instead of going out to HBM.  Needs N % 2048 == 0.
    int n_raw = blockIdx.x * RO_BLOCK + tid;
    if (A.xcd_remap) n_raw = 128 * ((int)(blockIdx.x & 7) + 16 * (int)(blockIdx.x >> 3) + 8 * (tid >> 7)) + (tid & 127);
    const bool valid = n_raw < A.N;
    const int n = valid ? n_raw : A.N - 1;
    const float4 *__restrict__ ap = A.a + n;
    const size_t stride = (size_t)A.N;

    // ---- issue the action stream first: one coalesced 1 KiB stripe per wave per step
    float4 ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = ap[(size_t)i * stride];
    __builtin_amdgcn_sched_barrier(0);  // keep hipcc from sinking the loads next to their first use

    // ---- wave-uniform horizon window held across lanes: lane k carries step k's targets
    // (free.py:150-155: targets = traj[time+1] after each step, gather clamps; step 0 = the state's own)
    float wpx, wpy, wpz, wvx, wvy, wvz, wdisc = 1.0f;
    {
        const int k = lane & (COVO_H - 1);
        int idx = time0 + k;
        idx = idx < 0 ? 0 : (idx > A.T - 1 ? A.T - 1 : idx);
        const bool own = (k == 0);
        wpx = own ? st[ST_POSTAR + 0] : A.pos_traj[3 * idx + 0];
        wpy = own ? st[ST_POSTAR + 1] : A.pos_traj[3 * idx + 1];
        wpz = own ? st[ST_POSTAR + 2] : A.pos_traj[3 * idx + 2];
        wvx = own ? st[ST_VELTAR + 0] : A.vel_traj[3 * idx + 0];
        wvy = own ? st[ST_VELTAR + 1] : A.vel_traj[3 * idx + 1];
        wvz = own ? st[ST_VELTAR + 2] : A.vel_traj[3 * idx + 2];
        if (!DISC1) {
            for (int i = 0; i < k; ++i) wdisc *= A.discount;  // discount^k (covo.py:258)
        }
    }
    const int kdone = A.max_steps - time0;  // steps k >= kdone see time >= max_steps (quadrotor.py:483)

    qm::State<float> s;
    s.px = st[ST_POS + 0]; s.py = st[ST_POS + 1]; s.pz = st[ST_POS + 2];
    s.vx = st[ST_VEL + 0]; s.vy = st[ST_VEL + 1]; s.vz = st[ST_VEL + 2];
    s.qx = st[ST_QUAT + 0]; s.qy = st[ST_QUAT + 1]; s.qz = st[ST_QUAT + 2]; s.qw = st[ST_QUAT + 3];
    s.ox = st[ST_OMEGA + 0]; s.oy = st[ST_OMEGA + 1]; s.oz = st[ST_OMEGA + 2];
    const float p0x = s.px, p0y = s.py, p0z = s.pz;
    const float f0x = st[ST_FDIST + 0], f0y = st[ST_FDIST + 1], f0z = st[ST_FDIST + 2];
    const float fsx = A.f_shared_dev ? A.f_shared_dev[0] : A.f_shared[0];
    const float fsy = A.f_shared_dev ? A.f_shared_dev[1] : A.f_shared[1];
    const float fsz = A.f_shared_dev ? A.f_shared_dev[2] : A.f_shared[2];
    const qm::Consts<float> c = A.c;

    float acc = 0.0f, r_before = 0.0f;   // covo.py:246-247
    bool done_before = false;

#pragma unroll
    for (int k = 0; k < COVO_H; ++k) {
        const float tx = lane_bcast(wpx, k), ty = lane_bcast(wpy, k), tz = lane_bcast(wpz, k);
        const float tvx = lane_bcast(wvx, k), tvy = lane_bcast(wvy, k), tvz = lane_bcast(wvz, k);
        // reward / termination of the PRE-step state (quadrotor.py:243-244)
        float r = qm::reward<float, float>(s, tx, ty, tz, tvx, tvy, tvz);
        const float pmax = fmaxf(fmaxf(fabsf(s.px), fabsf(s.py)), fabsf(s.pz));
        bool done = (k >= kdone) | (pmax > c.pos_limit);
        if (A.rollover)  // quadrotor.py:486-490
            done = done | (s.qw < 0.70710678118654752f) | (fmaxf(fmaxf(fabsf(s.ox), fabsf(s.oy)), fabsf(s.oz)) > 100.0f);
        r = done_before ? r_before : r;  // covo.py:233
        done_before = done_before | done;
        r_before = r;
        acc = DISC1 ? acc + r : fmaf(lane_bcast(wdisc, k), r, acc);  // covo.py:257-261

        float4 av = ring[k % PF];
        if (PF < COVO_H && k + PF < COVO_H) ring[k % PF] = ap[(size_t)(k + PF) * stride];
        if (CLIP) { av.x = qm::clip11_(av.x); av.y = qm::clip11_(av.y); av.z = qm::clip11_(av.z); av.w = qm::clip11_(av.w); }  // (lab baseline: no NaN variant)
        const float fx = (k == 0) ? f0x : fsx;
        const float fy = (k == 0) ? f0y : fsy;
        const float fz = (k == 0) ? f0z : fsz;
        qm::dyn_step<float, float>(s, av.x, av.y, av.z, av.w, c, fx, fy, fz);
        if (STATS) {  // covo.py:234-237: post-step positions, shifted by the initial position
            // No cross-lane reduction per step (6 butterfly sums x 32 steps = 192 dependent ds_bpermute chains per
            // wave made this variant 7x slower than the plain kernel): every lane parks its three offsets in a
            // wave-private LDS panel; after every 8 steps, lane p < 24 owns one (step, axis) column, walks its 64
            // entries (rotated by p: conflict-free banks) and forms sum and sum of squares in a fixed order.
            float *pan = spanel + wave * (8 * 3 * COVO_WAVE);
            const int kk = k & 7;
            pan[(kk * 3 + 0) * COVO_WAVE + lane] = valid ? s.px - p0x : 0.0f;
            pan[(kk * 3 + 1) * COVO_WAVE + lane] = valid ? s.py - p0y : 0.0f;
            pan[(kk * 3 + 2) * COVO_WAVE + lane] = valid ? s.pz - p0z : 0.0f;
            if (kk == 7) {
                if (lane < 24) {
                    const float *col = pan + lane * COVO_WAVE;  // lane = kk' * 3 + axis
                    double s1 = 0.0, s2 = 0.0;  // fp64: 64 identical offsets (step 0) must sum without rounding
#pragma unroll 8
                    for (int jj = 0; jj < COVO_WAVE; ++jj) {
                        const double v = (double)col[(jj + lane) & (COVO_WAVE - 1)];
                        s1 += v;
                        s2 = fma(v, v, s2);
                    }
                    const int ks = (k - 7) + lane / 3, ax = lane % 3;
                    double *sl = sacc + wave * (COVO_H * 6) + ks * 6;
                    sl[ax] = s1;
                    sl[3 + ax] = s2;
                }
            }
        }
    }
    const float cost = -acc;  // covo.py:263
    if (valid) A.cost[n] = cost;

    if (A.groupmin != nullptr) {
        const float wm = wave_min(valid ? cost : __builtin_inff());
        if (lane == 0 && (n_raw & ~(COVO_WAVE - 1)) < A.N) A.groupmin[n_raw >> 6] = wm;
    }
    if (A.records != nullptr)
        rollout_record<RO_BLOCK / COVO_WAVE, RO_BLOCK / COVO_WAVE>(A, cost, valid, n, wave, true, lane, blockIdx.x, rec_m, rec_s, rec_v);
    if (STATS) {
        __syncthreads();
        for (int i = tid; i < COVO_H * 6; i += RO_BLOCK)
            A.stats_ws[(size_t)blockIdx.x * (COVO_H * 6) + i] =
                (sacc[i] + sacc[COVO_H * 6 + i]) + (sacc[2 * COVO_H * 6 + i] + sacc[3 * COVO_H * 6 + i]);
    }
    RO_PROBE_END(0);
}

#endif  // ROLLOUT_LAB_BASELINE

#include "rollout_pipe.hpp"
