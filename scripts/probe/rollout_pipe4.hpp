// rollout_pipe4.hpp -- the N-sample x H-step rollout as a wave pipeline of 2 + CH stages (gfx950).  Round-4 experiment, NOT part of the library: included by scripts/probe/rollout_lab.hip only.
//
// Round 4.  What the machine does, calibrated with the counters on a known load (scripts/probe/valu_calib.hip, profiles/r04_valu_calib.json):
// a SIMD's VALU pipe takes one wave-instruction per 2.24 cycles when saturated, which needs >= 4 waves of dependent chains (8.6 /
// 4.3 / 2.9 / 2.4 cycles per instruction at 1 / 2 / 3 / 4 waves); ONE wave never issues faster than one instruction per ~5 cycles.
// At N = 65 536 a launch is one 64-sample group per SIMD, so the kernel's time is the instruction count of its LONGEST wave
// times ~4.5 cycles.  Round 2's three stages (attitude 29 + 2 LDS | translation 41 + 3 | reward 36 + 1 instructions per step) ran
// 44 on the critical wave at 3 waves per SIMD.  Here every 64 samples get 2 + CH waves (CH = 2: four), none longer than 31:
//
//   A  attitude (the serial chain)   stripes a[k][n] -> g = dt/2 omega (body-rate lag, free.py:105-107), q (x) (1, g), re-normalise
//                                    (free.py:96,104,139)                           -> ring A slot k: {x, y, z, w}, {tau | rollover flag}
//   C  translation chain             ring A -> termination of the PRE-step state (quadrotor.py:479-490) folded into the step's COST
//                                    COEFFICIENT (below), then v, p (free.py:92,97-103)  -> ring C slot k: {px, py, pz, coef}, {vx, vy, vz}
//   E_j, j < CH  evaluation          step CH c + j of every chunk c: ring A (q) + ring C -> errors, yaw terms, reward
//                                    (utils.py:266-313), acc_j += coef_k r_k          -> cost = -(acc_0 + acc_1 + ...)
//
// 29 + 2 | 22 + 4 | (55 + 3) / CH instructions per sample-step, one s_barrier per CH steps; A runs chunk i while C runs chunk
// i - 1 and the E waves chunk i - 2 (ring A is three chunks deep, ring C two).
//
// The done-freeze as a coefficient.  covo.py:233-263: r_eff_k = done_before ? r_before : r_k, cost = -sum_k d^k r_eff_k.  With j the
// first step whose PRE-step state is terminal, r_eff_k = r_k for k <= j and r_j for k > j, i.e.
//     cost = -sum_k coef_k r_k,   coef_k = d^k (k < j),   sum_{i >= j} d^i (k = j),   0 (k > j).
// Termination never needs the reward, so stage C -- which walks the steps in order anyway -- forms coef_k (two v_cndmask) and the
// evaluation of step k needs nothing from any other step: the E waves split the steps of a chunk among themselves.  (The sum's
// order differs from the sequential one at the 1-ulp level: per-wave partial sums, the frozen tail as ONE product.  Same 1e-5
// bar against the fp64 oracle as before, measured 1-2e-6.)  Everything else as in round 2's kernel (rollout_pipe.hpp keeps the
// notes on the arithmetic: re-normalisation at step 0 only, Q[2,2] = 1 - 2(x^2 + y^2), folded constants, |atan2| polynomial).
#pragma once

template <int CH>
struct Rp4Lds {
    float4 q[3 * CH][COVO_WAVE];    // ring A: x, y, z, w (unit)
    float tau[3 * CH][COVO_WAVE];   // ring A: thrust dt / m (sign bit: rollover flag of the stored state)
    float4 pc[2 * CH][COVO_WAVE];   // ring C: px, py, pz of the PRE-step state, coef_k
    float4 vv[2 * CH][COVO_WAVE];   // ring C: vx, vy, vz of the PRE-step state (.w unused; 12-byte stores)
};

__device__ __forceinline__ void rp4_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// ROLL, GROUPS, STATS, REWARD, FDIST, REC, BATCHED, DISC1: as rollout_pipe3_kernel (rollout_pipe.hpp).  CH: steps per barrier =
// evaluation waves per group; a workgroup is (2 + CH) GROUPS waves, stage-major (with CH = 2 and four groups every SIMD of the CU
// hosts exactly one A, one C and two E waves).
template <bool DISC1, bool ROLL, int CH, int GROUPS, bool BATCHED = false, bool STATS = false, bool REC = false, int REWARD = 0,
          int FDIST = 0>
__global__ __launch_bounds__((2 + CH) * GROUPS * COVO_WAVE) void rollout_pipe4_kernel(const RolloutArgs A_, const RolloutArgs *__restrict__ batch)
{
    static_assert(COVO_H % CH == 0, "CH must divide the horizon");
    constexpr int NCH = COVO_H / CH, NWG = (2 + CH) * GROUPS;
    const RolloutArgs &A = BATCHED ? batch[blockIdx.y] : A_;
    __shared__ Rp4Lds<CH> lds_all[GROUPS];
    __shared__ float lds_acc[GROUPS][CH > 1 ? CH - 1 : 1][COVO_WAVE];  // the partial sums of E_1 .. E_{CH-1}
    __shared__ float lds_st[STATS ? GROUPS : 1][STATS ? COVO_H : 1][9];  // STATS: per group and step {sum d, sum d^2, shift}
    const int lane = threadIdx.x & (COVO_WAVE - 1);
    const int wave_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave_ / GROUPS;  // 0 A, 1 C, 2 + j: E_j
    const int gsub = wave_ % GROUPS;
    Rp4Lds<CH> &lds = lds_all[gsub];
    const float *__restrict__ st = A.state;

    int group = blockIdx.x * GROUPS + gsub;  // XCD affinity (speed only), as in rollout_pipe3_kernel
    if (A.xcd_remap) {
        const int x = blockIdx.x & 7, m = (int)(blockIdx.x >> 3) * GROUPS + gsub, q = A.xcd_remap;
        group = q * (x + 8 * (m / q)) + (m % q);
    }
    const int n_raw = group * COVO_WAVE + lane;
    const bool valid = n_raw < A.N;
    const int n = valid ? n_raw : A.N - 1;
    const qm::Consts<float> c = A.c;

    if (role == 0) {
        // ============================================================ A: attitude
        const float4 *__restrict__ ap = A.a + n;
        const size_t stride = (size_t)A.N;
        constexpr int PF = 12 < COVO_H ? 12 : COVO_H;  // stripes in flight
        float4 ring[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) ring[i] = ap[(size_t)(i < COVO_H - 1 ? i : 0) * stride];
        const float ctau = c.thrust_half * c.inv_m * c.dt;  // tau = (a0 + 1) ctau = thrust dt / m
        const float kg0 = c.komega[0] * c.one_m_alpha * c.half_dt, kg1 = c.komega[1] * c.one_m_alpha * c.half_dt,
                    kg2 = c.komega[2] * c.one_m_alpha * c.half_dt;  // g' = alpha g + a kg,  g = dt/2 omega
        float x, y, z, w;
        {
            const float qx = st[ST_QUAT + 0], qy = st[ST_QUAT + 1], qz = st[ST_QUAT + 2], qw = st[ST_QUAT + 3];
            const float rn = qm::rsqrt_(qx * qx + qy * qy + qz * qz + qw * qw);  // free.py:88 (the noisy state is not unit)
            x = qx * rn; y = qy * rn; z = qz * rn; w = qw * rn;
        }
        float gx = st[ST_OMEGA + 0] * c.half_dt, gy = st[ST_OMEGA + 1] * c.half_dt, gz = st[ST_OMEGA + 2] * c.half_dt;
        const float groll = 100.0f * c.half_dt;
#pragma unroll
        for (int k = 0; k < COVO_H; ++k) {
            float4 a4 = ring[k % PF];
            if (k + PF < COVO_H - 1) ring[k % PF] = ap[(size_t)(k + PF) * stride];
            if (A.clip == 1) { a4.x = qm::clip11_(a4.x); a4.y = qm::clip11_(a4.y); a4.z = qm::clip11_(a4.z); a4.w = qm::clip11_(a4.w); }
            else if (A.clip == 2) {  // COVO_FLAG_PROPAGATE_NAN: jnp.clip's NaN semantics (quadrotor.py:223,258)
                a4.x = qm::clip11_nan_(a4.x); a4.y = qm::clip11_nan_(a4.y); a4.z = qm::clip11_nan_(a4.z); a4.w = qm::clip11_nan_(a4.w);
            }
            float tau = __builtin_fmaf(a4.x, ctau, ctau);  // quadrotor.py:259, free.py:82,98,103
            if (ROLL && k > 0) {  // quadrotor.py:486-490 on the stored state (step 0: wave C, from the state itself)
                const bool roll = (w < RP_COS_PI_4) | (fmaxf(fmaxf(fabsf(gx), fabsf(gy)), fabsf(gz)) > groll);
                tau = roll ? __int_as_float(__float_as_int(tau) | 0x80000000) : tau;
            }
            lds.q[k % (3 * CH)][lane] = make_float4(x, y, z, w);
            lds.tau[k % (3 * CH)][lane] = tau;
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
                rp4_barrier();
                rp_pin(x, y, z, w, gx, gy, gz);
            }
        }
        rp4_barrier();  // interval NCH: C on the last chunk
        rp4_barrier();  // interval NCH + 1: the E waves on the last chunk
        if (!REC && !STATS) return;
    }

    if (role == 1) {
        // ============================================================ C: translation chain + cost coefficients
        const int time0 = __float_as_int(st[ST_TIME]);
        const int kdone = A.max_steps - time0;  // steps k >= kdone see time >= max_steps (quadrotor.py:483)
        float px = st[ST_POS + 0], py = st[ST_POS + 1], pz = st[ST_POS + 2];
        float vx = st[ST_VEL + 0], vy = st[ST_VEL + 1], vz = st[ST_VEL + 2];
        // discount: lane k carries d^k and G_k = sum_{i >= k} d^i (explicit sums: exact integers at d = 1, so that the general path
        // equals the DISC1 one bit for bit there)
        float wdk = 1.0f, wgk = 0.0f;
        if (!DISC1) {
            const int kk = lane & (COVO_H - 1);
            float p = 1.0f;
            for (int i = 0; i < COVO_H; ++i) {
                if (i == kk) wdk = p;
                if (i >= kk) wgk += p;
                p *= A.discount;
            }
        }
        // STATS (covo.py:234-237, 281): see rollout_pipe.hpp (same reduction, on this wave now: it owns the positions)
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
        bool roll0 = false;
        if (ROLL) roll0 = (st[ST_QUAT + 3] < RP_COS_PI_4) |
                          (fmaxf(fmaxf(fabsf(st[ST_OMEGA + 0]), fabsf(st[ST_OMEGA + 1])), fabsf(st[ST_OMEGA + 2])) > 100.0f);
        bool frozen = false;  // a step before this one saw a terminal state (covo.py:233: done_before)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the table loads: rp4_barrier only waits on lgkmcnt
        rp4_barrier();  // interval 0: chunk 0 of ring A is being written
#pragma unroll
        for (int k = 0; k < COVO_H; ++k) {
            if (k % CH == 0) {
                rp_pin(px, py, pz, vx, vy, vz);
                if (FDIST == 2) rp_pin(fpx, fpy, fpz);
            }
            const float4 q4 = lds.q[k % (3 * CH)][lane];
            const float tau_raw = lds.tau[k % (3 * CH)][lane];
            const float x = q4.x, y = q4.y, z = q4.z, w = q4.w;
            // termination of the PRE-step state (quadrotor.py:479-490) -> this step's coefficient in the cost (see the header)
            const float pmax = fmaxf(fmaxf(fabsf(px), fabsf(py)), fabsf(pz));
            bool done = (k >= kdone) | (pmax > c.pos_limit);
            float tau = tau_raw;
            if (ROLL) {
                done = done | ((k == 0) ? roll0 : (__float_as_int(tau_raw) < 0));
                tau = fabsf(tau_raw);
            }
            const float dk = DISC1 ? 1.0f : lane_bcast(wdk, k), gk = DISC1 ? (float)(COVO_H - k) : lane_bcast(wgk, k);
            float coef = done ? gk : dk;
            coef = frozen ? 0.0f : coef;
            frozen = frozen | done;
            lds.pc[k % (2 * CH)][lane] = make_float4(px, py, pz, coef);
            *reinterpret_cast<float3 *>(&lds.vv[k % (2 * CH)][lane]) = make_float3(vx, vy, vz);
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
            if ((k + 1) % CH == 0) rp4_barrier();
        }
        rp4_barrier();  // interval NCH + 1: the E waves on the last chunk
        if (!REC && !STATS) return;
    }

    float cost = 0.0f;
    if (role >= 2) {
        // ============================================================ E_j: evaluation of step CH c + j of every chunk c
        const int j = role - 2;
        constexpr float LN2 = 0.69314718056f;
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
        float yn0, yd0;  // step 0: yaw terms of the un-normalised stored quaternion (utils.py:289-290); REWARD 1: yn0 = its w^2
        {
            const float qx = st[ST_QUAT + 0], qy = st[ST_QUAT + 1], qz = st[ST_QUAT + 2], qw = st[ST_QUAT + 3];
            yn0 = REWARD == 1 ? qw * qw : __builtin_fmaf(qw, qz, qx * qy);
            yd0 = __builtin_fmaf(-qz, qz, __builtin_fmaf(-qy, qy, 0.5f));
        }
        float acc = 0.0f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the window loads
        rp4_barrier();  // interval 0
        rp4_barrier();  // interval 1
#pragma unroll
        for (int ci = 0; ci < NCH; ++ci) {
            rp_pin(acc, wpx);
#pragma unroll
            for (int jj = 0; jj < CH; ++jj) {
                if (jj == j) {  // wave-uniform: this wave's step of the chunk (compile-time k inside)
                    const int k = ci * CH + jj;
                    const float4 q4 = lds.q[k % (3 * CH)][lane];
                    const float4 pc = lds.pc[k % (2 * CH)][lane];
                    const float3 v3 = *reinterpret_cast<const float3 *>(&lds.vv[k % (2 * CH)][lane]);
                    const float tx = lane_bcast(wpx, k), ty = lane_bcast(wpy, k), tz = lane_bcast(wpz, k);
                    const float dx = tx - pc.x, dy = ty - pc.y, dz = tz - pc.z;
                    const float ep2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                    float r;
                    if (REWARD == 1) {
                        // utils.py:297-313: r = -0.02 (5 err_pos^2 / 3 + 3 (1 - w^2)) = -(0.1 / 3) err_pos^2 + 0.06 w^2 - 0.06
                        const float w2 = (k == 0) ? yn0 : q4.w * q4.w;
                        r = __builtin_fmaf(ep2, -0.1f / 3.0f, __builtin_fmaf(w2, 0.06f, -0.06f));
                        (void)v3;
                    } else {
                        const float tvx = lane_bcast(wvx, k), tvy = lane_bcast(wvy, k), tvz = lane_bcast(wvz, k);
                        const float ex = tvx - v3.x, ey = tvy - v3.y, ez = tvz - v3.z;
                        const float ev2 = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
                        const float yn = (k == 0) ? yn0 : __builtin_fmaf(q4.w, q4.z, q4.x * q4.y);
                        const float yd = (k == 0) ? yd0 : __builtin_fmaf(-q4.z, q4.z, __builtin_fmaf(-q4.y, q4.y, 0.5f));
                        const float err_pos = qm::sqrt_(ep2), err_vel = qm::sqrt_(ev2);
                        const float yaw = rp3_atan2abs(yn, yd);
                        const float l2 = __builtin_amdgcn_logf(err_pos + 1.0f);  // log2
                        // utils.py:266-274, 285-294: r = 1.3 - 0.05 err_vel - (0.4 e + 0.4 sat(4 l) + 0.2 sat(8 l) + 0.1 sat(16 l) + 0.1 sat(32 l)) - 0.2 |yaw|
                        r = __builtin_fmaf(err_vel, -0.05f, 1.3f);
                        r = __builtin_fmaf(err_pos, -0.4f, r);
                        r = __builtin_fmaf(qm::sat01_(l2 * (4.0f * LN2)), -0.4f, r);
                        r = __builtin_fmaf(qm::sat01_(l2 * (8.0f * LN2)), -0.2f, r);
                        r = __builtin_fmaf(qm::sat01_(l2 * (16.0f * LN2)) + qm::sat01_(l2 * (32.0f * LN2)), -0.1f, r);
                        r = __builtin_fmaf(yaw, -0.2f, r);
                    }
                    acc = __builtin_fmaf(pc.w, r, acc);  // covo.py:233-263 through the coefficient
                }
            }
            if (ci == NCH - 1 && j > 0) lds_acc[gsub][j - 1][lane] = acc;  // before the last barrier: E_0 adds the partial sums
            rp4_barrier();
        }
        if (j == 0) {
#pragma unroll
            for (int jj = 1; jj < CH; ++jj) acc += lds_acc[gsub][jj - 1][lane];
            cost = -acc;  // covo.py:263
            if (valid) A.cost[n] = cost;
            if (A.groupmin != nullptr) {
                const float wm = wave_min(valid ? cost : __builtin_inff());
                if (lane == 0 && group * COVO_WAVE < A.N) A.groupmin[group] = wm;
            }
        } else if (!REC && !STATS) {
            return;
        }
    }
    if (REC) {  // every wave of the workgroup (only the E_0 waves carry a cost)
        __shared__ float rec_m[GROUPS], rec_s[GROUPS];
        __shared__ __attribute__((aligned(16))) float rec_v[GROUPS][COVO_NA];
        rollout_record<NWG, GROUPS>(A, cost, valid && role == 2, n, role == 2 ? gsub : 0, role == 2, lane, blockIdx.x, rec_m, rec_s,
                                    rec_v);
    }
    if (STATS) {
        // this workgroup's {sum (p - p0), sum (p - p0)^2} per step and axis, in fp64, with each C wave's shift put back:
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
                    const int x = blockIdx.x & 7, m = (int)(blockIdx.x >> 3) * GROUPS + g, qq = A.xcd_remap;
                    grp = qq * (x + 8 * (m / qq)) + (m % qq);
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
