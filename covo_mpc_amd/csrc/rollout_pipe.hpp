// rollout_pipe.hpp -- the N-sample x H-step rollout as a three-stage wave pipeline (gfx950).  Included by rollout.hip.
#pragma once

constexpr float RP_COS_PI_4 = 0.70710678118654752f;  // quadrotor.py:487

// Empty asm with the live state as in/out operands: volatile asms keep their order, so the arithmetic of a chunk can
// neither sink below the next barrier nor be hoisted above the previous one (LLVM's IR passes move pure arithmetic
// freely across the barrier asm otherwise -- observed: all barriers and ring reads first, 400 VGPRs).
__device__ __forceinline__ void rp_pin(float &a, float &b, float &c, float &d, float &e, float &f, float &g)
{
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g));
}
__device__ __forceinline__ void rp_pin(float &a, float &b, float &c, float &d, float &e, float &f)
{
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
__device__ __forceinline__ void rp_pin(float &a, float &b) { asm volatile("" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void rp_pin(float &a, float &b, float &c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); }

// Three-stage pipeline (the product kernel, with and without the position statistics of --info).  Measured on gfx950 (scripts/probe/valu_probe.hip, rollout_lab.hip): one wave
// retires a dependent VALU instruction per ~4 ns, a SIMD with three such waves one per ~1.2-1.5 ns, and every LDS
// instruction costs about as much SIMD time as 6-8 VALU instructions.  So a launch wants >= 3 waves per SIMD, as few VALU
// instructions as the arithmetic allows, and as few dwords through LDS as the stages can do with.  Every 64 samples get
// three waves that run the horizon one chunk of CH steps apart:
//
//   A (attitude)     stripes a[k][n] -> g = dt/2 omega (body-rate lag, free.py:105-107), quaternion step + re-normalise
//                    (free.py:96,104,139)                         -> ring A slot k: {x, y, z, w}, {tau = thrust dt / m}
//   T (translation)  ring A + per-step targets (v_readlane) -> squared position / velocity errors, termination flag and yaw
//                    terms of the PRE-step state (quadrotor.py:243-244, 479-490; utils.py:286-290)
//                                                                 -> ring T slot k: {err_pos^2, err_vel^2 | flag, yn, yd}
//                    then velocity and position (free.py:92,97-103)
//   R (reward)       ring T -> reward (utils.py:266-294), frozen reward, running cost (covo.py:233-263)
//
// ~28 + 41 + 34 VALU instructions per sample-step (the one-lane-per-sample kernel: 135), one s_barrier per CH steps.
// Differences to quad_model.hpp's operation order, all at the 1-ulp level (tests: <= 1e-5 relative on the cost against the
// fp64 oracle, as before): the stored quaternion of steps k >= 1 was normalised by the previous step (free.py:139), so
// free.py:88's re-normalisation is applied at step 0 only (it changes a unit quaternion by <= 1 ulp) and Q[2,2] is formed as
// 1 - 2(x^2 + y^2); dt/2, dt/m and (1 - alpha) are folded into the constants of their products; atan2's arguments are
// halved (exact); |atan| is a 6-term odd minimax polynomial (3.9e-7 abs).
#ifdef RP_TIMELINE
__device__ unsigned long long *g_rp_tl;  // scripts/probe/pipe_timeline.hip: [workgroup][3 stages][8] stamps
#define RP3_DECL() unsigned long long rp3_t_[4] = {0, 0, 0, 0}, rp3_c_[4] = {0, 0, 0, 0}
#define RP3_STAMP(I) do { rp3_t_[I] = wall_clock64(); rp3_c_[I] = __builtin_readcyclecounter(); } while (0)
#define RP3_FLUSH(ROLE) do { if (lane == 0 && g_rp_tl) { unsigned hw_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); \
    unsigned long long *o_ = g_rp_tl + ((size_t)blockIdx.x * 3 + (ROLE)) * 8; o_[0] = rp3_t_[0]; o_[1] = rp3_t_[1]; o_[2] = rp3_t_[2]; o_[3] = rp3_t_[3]; o_[4] = hw_; o_[5] = rp3_c_[3] - rp3_c_[1]; } } while (0)
#else
#define RP3_DECL()
#define RP3_STAMP(I)
#define RP3_FLUSH(ROLE)
#endif

template <int CH>
struct Rp3Lds {
    float4 q[2 * CH][COVO_WAVE];   // ring A: x, y, z, w (unit)
    float tau[2 * CH][COVO_WAVE];  // ring A: thrust dt / m (sign bit: rollover flag of the stored state)
    float4 t[2 * CH][COVO_WAVE];   // ring T: err_pos^2, err_vel^2 (sign bit: terminated), yn/2, yd/2 of the stored quaternion
};

template <int ONLY>
__device__ __forceinline__ void rp3_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    if (ONLY == -1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// |atan2(y, x)|: odd minimax polynomial of atan on [0,1] (6 terms, 3.9e-7 abs) + octant fix-up; atan2(0, 0) = 0 through the
// floor on the larger magnitude.  17 VALU.
__device__ __forceinline__ float rp3_atan2abs(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = __builtin_fmaxf(__builtin_fmaxf(ax, ay), 1e-30f);
    const float mn = __builtin_amdgcn_fmed3f(ax, ay, 0.0f);  // min of two non-negative numbers as ONE v_med3 with |.| modifiers
    const float t = mn * __builtin_amdgcn_rcpf(mx);
    const float s = t * t;
    float r = 0.007374854292720556f;
    r = __builtin_fmaf(r, s, -0.03552231565117836f);
    r = __builtin_fmaf(r, s, 0.08217037469148636f);
    r = __builtin_fmaf(r, s, -0.13398927450180054f);
    r = __builtin_fmaf(r, s, 0.1986188441514969f);
    r = __builtin_fmaf(r, s, -0.33325397968292236f);
    r = r * s;
    r = __builtin_fmaf(r, t, t);
    r = (ay > ax) ? (1.57079637f - r) : r;
    r = (x < 0.0f) ? (3.14159274f - r) : r;
    return r;
}

// ROLL: the rollover termination of is_terminal (quadrotor.py:486-490) is on (Quad3D(disable_rollover_terminate=False)).
// GROUPS: 64-sample groups per workgroup (3 GROUPS waves, stage-major: waves [0, GROUPS) are the A waves, ... -- with four
//   groups every SIMD of the CU hosts one wave of each stage); CH: steps per barrier.
// ONLY >= 0 (scripts/probe/rollout_lab.hip): every wave runs stage ONLY on its own LDS copy, no barriers -- the stage's
//   instruction stream in isolation; ONLY == -2: all three stages without barriers (timing bound, garbage results).
// STATS: the per-step position sums of covo.py:281 (pos_mean / pos_std) are formed by the T waves (see there).
// REWARD: env.reward_fn -- 0 tracking_penyaw_reward_fn (utils.py:285-294), 1 tracking_realworld_reward_fn (utils.py:297-313:
//   ring T then carries {err_pos^2, done flag as +-1, w^2 of the stored quaternion}).
// FDIST: how the disturbance of steps k >= 1 (free.py:147) reaches stage T -- 0: ONE wave-uniform vector for all of them
//   (none; gaussian from the shared step key); 1: a wave-uniform vector per step from A.f_tab (periodic, sin: lane k carries
//   step k's, v_readlane like the targets: 3 instructions per step, nothing per sample); 2: a per-sample force
//   f_k = A.drag_k rel |rel| + c_k f_{k-1} + g_k, rel = vel_{k-1} - A.drag_off (drag, mixed = (drag + sin + periodic) / 3:
//   free.py:41-56; rows {g_k, c_k} of A.f_tab, disturb.hip).
// REC: every workgroup also leaves its online-softmax record (rollout_record; A.records != null) -- the variant the fused
//   step runs; a template argument so that the profiler lists it under its own name.
// The three stages as a device function: the caller (rollout_pipe3_kernel below; step_small.hip's fused launch) owns the LDS and
// decides which wave plays which role for which 64-sample group.  A_LDS: stage A takes the action stripes from `a_lds`
// ([H][64] float4 of THIS group, left there by the launch's own noise draw) instead of A.a in HBM.  Returns this lane's cost
// (role 2), its sample index n and whether the sample exists.  KEEP_ALL: the A and T waves come back too (the caller has an
// epilogue for every wave: softmax records, position statistics); otherwise they leave the launch when their stage is done.
// KSTEPS (scripts/probe/rollout_lab.hip only): the stages stop after KSTEPS steps -- the timing bound of a split-horizon launch.
template <bool DISC1, bool ROLL, int CH, int ONLY, bool STATS, bool KEEP_ALL, int REWARD, int FDIST, bool A_LDS, class StatsLds,
          int KSTEPS = COVO_H>
__device__ __forceinline__ void rp3_stages(const RolloutArgs &A, Rp3Lds<CH> &lds, StatsLds &lds_st, const int role, const int gsub,
                                           const int group, const int lane, const float4 *__restrict__ a_lds, float &cost, bool &valid_out,
                                           int &n_out)
{
    static_assert(COVO_H % CH == 0, "CH must divide the horizon");
    RP3_DECL();
    const float *__restrict__ st = A.state;
    const int n_raw = group * COVO_WAVE + lane;
    const bool valid = n_raw < A.N;
    const int n = valid ? n_raw : A.N - 1;
    valid_out = valid;
    n_out = n;
    const qm::Consts<float> c = A.c;

    if (role == 0) {
        // ============================================================ A: attitude (the serial chain: runs at high priority)
        const float4 *__restrict__ ap = A.a + n;
        const size_t stride = (size_t)A.N;
        constexpr int PF = A_LDS ? 1 : (12 < COVO_H ? 12 : COVO_H);  // stripes in flight
        float4 ring[PF];
        RP3_STAMP(0);
        if (!A_LDS) {
#pragma unroll
            for (int i = 0; i < PF; ++i) ring[i] = ap[(size_t)(i < COVO_H - 1 ? i : 0) * stride];
        }
        const float ctau = c.thrust_half * c.inv_m * c.dt;                     // tau = (a0 + 1) ctau = thrust dt / m
        const float kg0 = c.komega[0] * c.one_m_alpha * c.half_dt, kg1 = c.komega[1] * c.one_m_alpha * c.half_dt,
                    kg2 = c.komega[2] * c.one_m_alpha * c.half_dt;            // g' = alpha g + a kg,  g = dt/2 omega
        float x, y, z, w;
        {
            const float qx = st[ST_QUAT + 0], qy = st[ST_QUAT + 1], qz = st[ST_QUAT + 2], qw = st[ST_QUAT + 3];
            const float rn = qm::rsqrt_(qx * qx + qy * qy + qz * qz + qw * qw);  // free.py:88 (the noisy state is not unit)
            x = qx * rn; y = qy * rn; z = qz * rn; w = qw * rn;
        }
        float gx = st[ST_OMEGA + 0] * c.half_dt, gy = st[ST_OMEGA + 1] * c.half_dt, gz = st[ST_OMEGA + 2] * c.half_dt;
        const float groll = 100.0f * c.half_dt;
#pragma unroll
        for (int k = 0; k < KSTEPS; ++k) {
            float4 a4;
            if (A_LDS) {
                a4 = a_lds[k * COVO_WAVE + lane];
            } else {
                a4 = ring[k % PF];
                if (k + PF < COVO_H - 1) ring[k % PF] = ap[(size_t)(k + PF) * stride];
            }
            if (A.clip == 1) { a4.x = qm::clip11_(a4.x); a4.y = qm::clip11_(a4.y); a4.z = qm::clip11_(a4.z); a4.w = qm::clip11_(a4.w); }
            else if (A.clip == 2) {  // COVO_FLAG_PROPAGATE_NAN: jnp.clip's NaN semantics (quadrotor.py:223,258)
                a4.x = qm::clip11_nan_(a4.x); a4.y = qm::clip11_nan_(a4.y); a4.z = qm::clip11_nan_(a4.z); a4.w = qm::clip11_nan_(a4.w);
            }
            float tau = __builtin_fmaf(a4.x, ctau, ctau);  // quadrotor.py:259, free.py:82,98,103
            if (ROLL && k > 0) {  // quadrotor.py:486-490 on the stored state (step 0: wave T, from the state itself)
                const bool roll = (w < RP_COS_PI_4) | (fmaxf(fmaxf(fabsf(gx), fabsf(gy)), fabsf(gz)) > groll);
                tau = roll ? __int_as_float(__float_as_int(tau) | 0x80000000) : tau;
            }
            lds.q[k % (2 * CH)][lane] = make_float4(x, y, z, w);
            lds.tau[k % (2 * CH)][lane] = tau;
            if (k < COVO_H - 1) {
                // q + dt/2 L(q) H omega (free.py:96,104) = q (x) (1, g)
                const float nx = __builtin_fmaf(-z, gy, __builtin_fmaf(y, gz, __builtin_fmaf(w, gx, x)));
                const float ny = __builtin_fmaf(-x, gz, __builtin_fmaf(z, gx, __builtin_fmaf(w, gy, y)));
                const float nz = __builtin_fmaf(-y, gx, __builtin_fmaf(x, gy, __builtin_fmaf(w, gz, z)));
                const float nw = __builtin_fmaf(-z, gz, __builtin_fmaf(-y, gy, __builtin_fmaf(-x, gx, w)));
                gx = __builtin_fmaf(gx, c.alpha, a4.y * kg0);  // free.py:105-107, 122
                gy = __builtin_fmaf(gy, c.alpha, a4.z * kg1);
                gz = __builtin_fmaf(gz, c.alpha, a4.w * kg2);
                const float rn = qm::rsqrt_(__builtin_fmaf(nw, nw, __builtin_fmaf(nz, nz, __builtin_fmaf(ny, ny, nx * nx))));  // free.py:139
                x = nx * rn; y = ny * rn; z = nz * rn; w = nw * rn;
            }
            if ((k + 1) % CH == 0) {
                if (k + 1 == CH) RP3_STAMP(1);
                if (k + 1 == COVO_H / 2) RP3_STAMP(2);
                if (k + 1 == COVO_H) RP3_STAMP(3);
                rp3_barrier<ONLY>();
                rp_pin(x, y, z, w, gx, gy, gz);
            }
        }
        RP3_FLUSH(0);
        rp3_barrier<ONLY>();
        rp3_barrier<ONLY>();
        if (!KEEP_ALL) return;
    }

    if (role == 1) {
        // ============================================================ T: translation
        const int time0 = __float_as_int(st[ST_TIME]);
        // wave-uniform horizon window held across lanes: lane k carries step k's targets
        // (free.py:150-155: targets = traj[time+1] after each step, gather clamps; step 0 = the state's own)
        float wpx, wpy, wpz, wvx, wvy, wvz;
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
        }
        const int kdone = A.max_steps - time0;  // steps k >= kdone see time >= max_steps (quadrotor.py:483)
        float px = st[ST_POS + 0], py = st[ST_POS + 1], pz = st[ST_POS + 2];
        float vx = st[ST_VEL + 0], vy = st[ST_VEL + 1], vz = st[ST_VEL + 2];
        // STATS (covo.py:234-237, 281: mean / std over the samples of every step's NEW position).  Per step the wave reduces six
        // numbers -- d and d^2 per axis, d = position minus the position of the wave's FIRST sample (a v_readlane; wave-uniform, so
        // coinciding samples sum exact zeros whatever their distance from the start and the fp32 squares stay as small as the
        // spread; the workgroup puts the shift back in fp64) -- in 19 instructions: v_permlane32_swap / v_permlane16_swap fold
        // two quantities at a time into the halves / rows of ONE register (6 -> 3 -> 2 registers), four DPP row rotations finish
        // both; the row sums and the shift are parked in lane (k & 15) of their row and written to LDS every 16 steps.  (One
        // butterfly per quantity: 6 x 8; round 1's one-lane kernel parked the offsets in LDS and had 24 lanes walk them in fp64
        // every 8 steps: 27.6 us at N = 65 536.)
        float st_a = 0.0f, st_b = 0.0f, st_c = 0.0f;
        auto stats_step = [&](int k, float nx, float ny, float nz) {
            const float cx = lane_bcast(nx, 0), cy = lane_bcast(ny, 0), cz = lane_bcast(nz, 0);
            float d0 = nx - cx, d1 = ny - cy, d2 = nz - cz;
            d0 = valid ? d0 : 0.0f;
            d1 = valid ? d1 : 0.0f;
            d2 = valid ? d2 : 0.0f;
            auto fold32 = [](float a, float b) {  // lanes < 32: a[l] + a[l + 32]; lanes >= 32: b[l - 32] + b[l]
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
                return __uint_as_float(r[0]) + __uint_as_float(r[1]);
            };
            auto fold16 = [](float a, float b) {  // rows 0 / 2: a's row pair sums; rows 1 / 3: b's
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
                return __uint_as_float(r[0]) + __uint_as_float(r[1]);
            };
            const float r0 = fold32(d0, d1), r1 = fold32(d2, d0 * d0), r2 = fold32(d1 * d1, d2 * d2);
            float sa = fold16(r0, r1);  // rows: sum d0 | sum d2 | sum d1 | sum d0^2   (4 samples per lane)
            float sb = fold16(r2, r2);  // rows: sum d1^2 | (same) | sum d2^2 | (same)
            sa += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sa), 0x128, 0xf, 0xf, false));
            sb += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sb), 0x128, 0xf, 0xf, false));
            sa += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sa), 0x124, 0xf, 0xf, false));
            sb += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sb), 0x124, 0xf, 0xf, false));
            sa += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sa), 0x122, 0xf, 0xf, false));
            sb += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sb), 0x122, 0xf, 0xf, false));
            sa += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sa), 0x121, 0xf, 0xf, false));
            sb += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sb), 0x121, 0xf, 0xf, false));
            const bool mine = (lane & 15) == (k & 15);
            const int row = lane >> 4;
            st_a = mine ? sa : st_a;
            st_b = mine ? sb : st_b;
            st_c = mine ? (row == 0 ? cx : (row == 1 ? cy : cz)) : st_c;
            if ((k & 15) == 15) {
                const int step = (k - 15) + (lane & 15);
                const int qa = row == 0 ? 0 : (row == 1 ? 2 : (row == 2 ? 1 : 3));
                lds_st[gsub][step][qa] = st_a;
                if ((row & 1) == 0) lds_st[gsub][step][row == 0 ? 4 : 5] = st_b;
                if (row < 3) lds_st[gsub][step][6 + row] = st_c;
            }
        };
        const float kf = c.inv_m * c.dt;  // v += dt/m f (free.py:98,103)
        const float gdt = c.neg_g * c.dt;
        const float c0x = st[ST_FDIST + 0] * kf, c0y = st[ST_FDIST + 1] * kf, c0z = __builtin_fmaf(st[ST_FDIST + 2], kf, gdt);
        float csx = 0.0f, csy = 0.0f, csz = 0.0f;       // FDIST 0: dt/m f_shared (+ dt g) for every step k >= 1
        float wfx = 0.0f, wfy = 0.0f, wfz = 0.0f, wfc = 0.0f;  // FDIST 1 / 2: lane k carries row k of the table
        float fpx = st[ST_FDIST + 0], fpy = st[ST_FDIST + 1], fpz = st[ST_FDIST + 2];  // FDIST 2: this sample's force
        if (FDIST == 0) {
            const float fsx = A.f_shared_dev ? A.f_shared_dev[0] : A.f_shared[0];
            const float fsy = A.f_shared_dev ? A.f_shared_dev[1] : A.f_shared[1];
            const float fsz = A.f_shared_dev ? A.f_shared_dev[2] : A.f_shared[2];
            csx = fsx * kf; csy = fsy * kf; csz = __builtin_fmaf(fsz, kf, gdt);
        } else {
            const float4 row = A.f_tab[lane & (COVO_H - 1)];
            if (FDIST == 1) { wfx = row.x * kf; wfy = row.y * kf; wfz = __builtin_fmaf(row.z, kf, gdt); }
            else { wfx = row.x; wfy = row.y; wfz = row.z; wfc = row.w; }
        }
        float yn0, yd0;  // step 0: yaw terms of the un-normalised stored quaternion (utils.py:289-290); REWARD 1: yn0 = its w^2
        bool roll0 = false;
        {
            const float qx = st[ST_QUAT + 0], qy = st[ST_QUAT + 1], qz = st[ST_QUAT + 2], qw = st[ST_QUAT + 3];
            yn0 = REWARD == 1 ? qw * qw : __builtin_fmaf(qw, qz, qx * qy);
            yd0 = __builtin_fmaf(-qz, qz, __builtin_fmaf(-qy, qy, 0.5f));
            if (ROLL) roll0 = (qw < RP_COS_PI_4) |
                              (fmaxf(fmaxf(fabsf(st[ST_OMEGA + 0]), fabsf(st[ST_OMEGA + 1])), fabsf(st[ST_OMEGA + 2])) > 100.0f);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the window loads: rp3_barrier only waits on lgkmcnt
        RP3_STAMP(0);
        rp3_barrier<ONLY>();  // interval 0: chunk 0 of ring A is being written
#pragma unroll
        for (int k = 0; k < KSTEPS; ++k) {
            if (k % CH == 0) {
                rp_pin(px, py, pz, vx, vy, vz);
                if (FDIST == 2) rp_pin(fpx, fpy, fpz);
            }
            const float4 q4 = lds.q[k % (2 * CH)][lane];
            const float tau_raw = lds.tau[k % (2 * CH)][lane];
            const float tx = lane_bcast(wpx, k), ty = lane_bcast(wpy, k), tz = lane_bcast(wpz, k);
            const float tvx = lane_bcast(wvx, k), tvy = lane_bcast(wvy, k), tvz = lane_bcast(wvz, k);
            const float x = q4.x, y = q4.y, z = q4.z, w = q4.w;
            // squared errors, termination and yaw terms of the PRE-step state (quadrotor.py:243-244, 479-490; utils.py:286-290)
            const float dx = tx - px, dy = ty - py, dz = tz - pz;
            const float ex = tvx - vx, ey = tvy - vy, ez = tvz - vz;
            const float ep2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            float ev2 = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
            const float pmax = fmaxf(fmaxf(fabsf(px), fabsf(py)), fabsf(pz));
            bool done = (k >= kdone) | (pmax > c.pos_limit);
            float tau = tau_raw;
            if (ROLL) {
                done = done | ((k == 0) ? roll0 : (__float_as_int(tau_raw) < 0));
                tau = fabsf(tau_raw);
            }
            if (REWARD == 1) {  // utils.py:297-313 needs err_pos^2 and the stored w^2 only; .y carries the termination flag
                lds.t[k % (2 * CH)][lane] = make_float4(ep2, done ? -1.0f : 1.0f, (k == 0) ? yn0 : w * w, 0.0f);
            } else {
                ev2 = done ? -ev2 : ev2;
                const float yn = (k == 0) ? yn0 : __builtin_fmaf(w, z, x * y);
                const float yd = (k == 0) ? yd0 : __builtin_fmaf(-z, z, __builtin_fmaf(-y, y, 0.5f));
                lds.t[k % (2 * CH)][lane] = make_float4(ep2, ev2, yn, yd);
            }
            if (k < COVO_H - 1) {
                // Q[:,2] of qtoQ(q) (geom.py:68-77) for a unit quaternion: 2 (xz + yw), 2 (yz - xw), 1 - 2 (x^2 + y^2);
                // v += dt (Q [0,0,T] + f)/m + dt [0,0,-g] (free.py:97-99,103); position with the OLD velocity (free.py:102)
                const float tau2 = tau + tau;
                const float u0 = __builtin_fmaf(x, z, y * w), u1 = __builtin_fmaf(y, z, -(x * w));
                const float s2 = __builtin_fmaf(y, y, x * x);
                px = __builtin_fmaf(vx, c.dt, px);
                py = __builtin_fmaf(vy, c.dt, py);
                pz = __builtin_fmaf(vz, c.dt, pz);
                if (FDIST == 2) {
                    // the NEXT step's force from this step's PRE-step velocity (free.py:147,41-56), then this step's velocity
                    // with the force it already carries (free.py:98,103)
                    const float rx = vx - A.drag_off[0], ry = vy - A.drag_off[1], rz = vz - A.drag_off[2];
                    const int kn = (k + 1) & (COVO_H - 1);
                    const float cn = lane_bcast(wfc, kn);
                    const float nfx = __builtin_fmaf(rx * __builtin_fabsf(rx), A.drag_k, __builtin_fmaf(cn, fpx, lane_bcast(wfx, kn)));
                    const float nfy = __builtin_fmaf(ry * __builtin_fabsf(ry), A.drag_k, __builtin_fmaf(cn, fpy, lane_bcast(wfy, kn)));
                    const float nfz = __builtin_fmaf(rz * __builtin_fabsf(rz), A.drag_k, __builtin_fmaf(cn, fpz, lane_bcast(wfz, kn)));
                    vx = __builtin_fmaf(fpx, kf, __builtin_fmaf(u0, tau2, vx));
                    vy = __builtin_fmaf(fpy, kf, __builtin_fmaf(u1, tau2, vy));
                    vz = __builtin_fmaf(fpz, kf, __builtin_fmaf(-tau2, s2, vz + tau)) + gdt;
                    fpx = nfx; fpy = nfy; fpz = nfz;
                } else {
                    vx = __builtin_fmaf(u0, tau2, vx);
                    vy = __builtin_fmaf(u1, tau2, vy);
                    vz = __builtin_fmaf(-tau2, s2, vz + tau);
                    if (k == 0) { vx += c0x; vy += c0y; vz += c0z; }
                    else if (FDIST == 1) { vx += lane_bcast(wfx, k); vy += lane_bcast(wfy, k); vz += lane_bcast(wfz, k); }
                    else { vx += csx; vy += csy; vz += csz; }
                }
                if (STATS) stats_step(k, px, py, pz);
            } else if (STATS) {  // the last step's new position enters no cost, only the statistics
                stats_step(k, __builtin_fmaf(vx, c.dt, px), __builtin_fmaf(vy, c.dt, py), __builtin_fmaf(vz, c.dt, pz));
            }
            if ((k + 1) % CH == 0) {
                if (k + 1 == CH) RP3_STAMP(1);
                if (k + 1 == COVO_H / 2) RP3_STAMP(2);
                if (k + 1 == COVO_H) RP3_STAMP(3);
                rp3_barrier<ONLY>();
            }
        }
        RP3_FLUSH(1);
        rp3_barrier<ONLY>();
        if (!KEEP_ALL) return;
    }

    cost = 0.0f;
    if (role == 2) {
    // ================================================================ R: reward
    constexpr float LN2 = 0.69314718056f;
    float acc = 0.0f, r_before = 0.0f, dk = 1.0f;  // covo.py:246-247
    bool done_before = false;
    RP3_STAMP(0);
    rp3_barrier<ONLY>();
    rp3_barrier<ONLY>();
#pragma unroll
    for (int k = 0; k < KSTEPS; ++k) {
        if (k % CH == 0) {
            rp_pin(acc, r_before);
        }
        const float4 e4 = lds.t[k % (2 * CH)][lane];
        float r;
        if (REWARD == 1) {
            // utils.py:297-313: r = -0.02 (5 err_pos^2 / 3 + 3 (1 - w^2)) = -(0.1 / 3) err_pos^2 + 0.06 w^2 - 0.06
            r = __builtin_fmaf(e4.x, -0.1f / 3.0f, __builtin_fmaf(e4.z, 0.06f, -0.06f));
        } else {
            const float err_pos = qm::sqrt_(e4.x), err_vel = qm::sqrt_(fabsf(e4.y));
            const float yaw = rp3_atan2abs(e4.z, e4.w);
            const float l2 = __builtin_amdgcn_logf(err_pos + 1.0f);  // log2
            // utils.py:266-274, 285-294: r = 1.3 - 0.05 err_vel - (0.4 e + 0.4 sat(4 l) + 0.2 sat(8 l) + 0.1 sat(16 l) + 0.1 sat(32 l)) - 0.2 |yaw|
            r = __builtin_fmaf(err_vel, -0.05f, 1.3f);
            r = __builtin_fmaf(err_pos, -0.4f, r);
            r = __builtin_fmaf(qm::sat01_(l2 * (4.0f * LN2)), -0.4f, r);
            r = __builtin_fmaf(qm::sat01_(l2 * (8.0f * LN2)), -0.2f, r);
            r = __builtin_fmaf(qm::sat01_(l2 * (16.0f * LN2)) + qm::sat01_(l2 * (32.0f * LN2)), -0.1f, r);
            r = __builtin_fmaf(yaw, -0.2f, r);
        }
        const bool done = __float_as_int(e4.y) < 0;
        r = done_before ? r_before : r;  // covo.py:233
        done_before = done_before | done;
        r_before = r;
        if (DISC1) acc += r;
        else { acc = __builtin_fmaf(dk, r, acc); dk *= A.discount; }  // covo.py:257-261
        if (k + 1 == CH) RP3_STAMP(1);
        if (k + 1 == COVO_H / 2) RP3_STAMP(2);
        if (k + 1 == COVO_H) RP3_STAMP(3);
        if ((k + 1) % CH == 0) rp3_barrier<ONLY>();  // (the last one only keeps the three stages' barrier counts equal)
    }
    RP3_FLUSH(2);
    cost = -acc;  // covo.py:263
    if (valid) A.cost[n] = cost;
    if (A.groupmin != nullptr) {
        const float wm = wave_min(valid ? cost : __builtin_inff());
        if (lane == 0 && group * COVO_WAVE < A.N) A.groupmin[group] = wm;
    }
    }
}

// the s_barrier count of one stage wave (COVO_H / CH + 2): what a wave of the workgroup that plays no stage executes next to them
template <int CH>
__device__ __forceinline__ void rp3_idle_barriers()
{
#pragma unroll 1
    for (int i = 0; i < COVO_H / CH + 2; ++i) rp3_barrier<-1>();
}

template <bool DISC1, bool ROLL, int CH, int GROUPS, bool BATCHED = false, int ONLY = -1, int ONLY_WAVES = 3, bool STATS = false,
          bool REC = false, int REWARD = 0, int FDIST = 0>
__global__ __launch_bounds__((ONLY >= 0 ? ONLY_WAVES : 3 * GROUPS) * COVO_WAVE) void rollout_pipe3_kernel(
    const RolloutArgs A_, const RolloutArgs *__restrict__ batch)
{
    // BATCHED: instance y's argument block is loaded from memory; A_ is instance 0's block as a kernel argument.  The pointers
    // are re-expressed relative to A_'s (covo_common.hpp: rebase_global -- flat accesses would tie the stripe prefetch to the LDS
    // rings' lgkmcnt waits); all instances alike in which nullable buffers they pass
    RolloutArgs Ab;
    if (BATCHED) {
        Ab = batch[blockIdx.y];
        Ab.state = rebase_global(A_.state, Ab.state);
        Ab.pos_traj = rebase_global(A_.pos_traj, Ab.pos_traj);
        Ab.vel_traj = rebase_global(A_.vel_traj, Ab.vel_traj);
        Ab.a = rebase_global(A_.a, Ab.a);
        Ab.cost = rebase_global(A_.cost, Ab.cost);
        Ab.groupmin = rebase_global(A_.groupmin, Ab.groupmin);
        Ab.stats_ws = rebase_global(A_.stats_ws, Ab.stats_ws);
        Ab.f_shared_dev = rebase_global(A_.f_shared_dev, Ab.f_shared_dev);
        Ab.f_tab = rebase_global(A_.f_tab, Ab.f_tab);
        Ab.records = rebase_global(A_.records, Ab.records);
    }
    const RolloutArgs &A = BATCHED ? Ab : A_;
    __shared__ Rp3Lds<CH> lds_all[ONLY >= 0 ? ONLY_WAVES : GROUPS];
    __shared__ float lds_st[STATS ? GROUPS : 1][STATS ? COVO_H : 1][9];  // STATS: per group and step {sum d, sum d^2, shift} (d: see stage T)
    const int lane = threadIdx.x & (COVO_WAVE - 1);
    const int wave_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = ONLY >= 0 ? ONLY : wave_ / GROUPS;  // role-major: the GROUPS waves of one stage are neighbours
    const int gsub = ONLY >= 0 ? 0 : wave_ % GROUPS;
    Rp3Lds<CH> &lds = lds_all[ONLY >= 0 ? wave_ : gsub];
    const float *__restrict__ st = A.state;

    int group = blockIdx.x * GROUPS + gsub;  // XCD affinity as in the two-stage kernel (speed only)
    if (A.xcd_remap) {  // q = 64-sample groups per GEMM workgroup: group = q * (GEMM workgroup on this XCD) + (its m % q-th group)
        const int x = blockIdx.x & 7, m = (int)(blockIdx.x >> 3) * GROUPS + gsub, q = A.xcd_remap;
        group = q * (x + 8 * (m / q)) + (m % q);
    }
    float cost = 0.0f;
    bool valid = false;
    int n = 0;
    rp3_stages<DISC1, ROLL, CH, ONLY, STATS, REC || STATS, REWARD, FDIST, false>(A, lds, lds_st, role, gsub, group, lane, nullptr, cost,
                                                                                  valid, n);
    if (ONLY == -1 && REC) {  // every wave of the workgroup (the A and T waves carry no cost)
        __shared__ float rec_m[GROUPS], rec_s[GROUPS];
        __shared__ __attribute__((aligned(16))) float rec_v[GROUPS][COVO_NA];
        rollout_record<3 * GROUPS, GROUPS>(A, cost, valid && role == 2, n, role == 2 ? gsub : 0, role == 2, lane, blockIdx.x, rec_m,
                                           rec_s, rec_v);
    }
    if (STATS) {
        // this workgroup's {sum (p - p0), sum (p - p0)^2} per step and axis, in fp64, with each T wave's shift put back:
        // p - p0 = d + D, D = (the wave's first sample) - p0  ->  sum = S1 + n D,  sum of squares = S2 + 2 D S1 + n D^2
        __syncthreads();
        const int t = threadIdx.x;
        if (t < COVO_H * 6) {
            const int k = t / 6, q = t % 6, ax = q % 3;
            const float p0 = st[ST_POS + ax];
            double tot = 0.0;
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
                int grp = blockIdx.x * GROUPS + g;
                if (A.xcd_remap) {
                    const int x = blockIdx.x & 7, m = (int)(blockIdx.x >> 3) * GROUPS + g, q = A.xcd_remap;
                    grp = q * (x + 8 * (m / q)) + (m % q);
                }
                int nv = A.N - grp * COVO_WAVE;
                nv = nv < 0 ? 0 : (nv > COVO_WAVE ? COVO_WAVE : nv);
                const double s1 = (double)lds_st[g][k][ax];
                const double D = (double)lds_st[g][k][6 + ax] - (double)p0;
                if (q < 3) tot += s1 + (double)nv * D;
                else tot += (double)lds_st[g][k][q] + 2.0 * D * s1 + (double)nv * D * D;
            }
            A.stats_ws[(size_t)blockIdx.x * (COVO_H * 6) + t] = tot;
        }
    }
}
