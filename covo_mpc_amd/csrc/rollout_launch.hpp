// rollout_launch.hpp -- launch helpers of the pipelined rollout kernel, shared by the translation units that instantiate it.
#pragma once
#include "rollout_common.hpp"

// 64-sample groups per workgroup of the pipelined kernel: as few as keep the workgroup count within the 256 records the
// merge takes in one pass, four (one wave of each stage on every SIMD) once the launch has >= 1024 groups to spread
static inline int pipe_groups(int N, int nbatch)
{
    const int ng = (N + COVO_WAVE - 1) / COVO_WAVE;
    if ((long long)ng * nbatch >= 1024 || ng > 512) return 4;
    return ng > 256 ? 2 : 1;
}

template <bool DISC1, bool ROLL, bool BATCHED, bool STATS, bool REC, int REWARD, int FDIST>
static void launch_pipe3_groups(const RolloutArgs &A, const RolloutArgs *batch, int nb, int groups, hipStream_t s)
{
    const int ng = (A.N + COVO_WAVE - 1) / COVO_WAVE;
    const dim3 grid((ng + groups - 1) / groups, nb);
    if (groups == 4)
        hipLaunchKernelGGL((rollout_pipe3_kernel<DISC1, ROLL, 2, 4, BATCHED, -1, 3, STATS, REC, REWARD, FDIST>), grid, dim3(3 * 4 * COVO_WAVE), 0, s, A, batch);
    else if (groups == 2)
        hipLaunchKernelGGL((rollout_pipe3_kernel<DISC1, ROLL, 2, 2, BATCHED, -1, 3, STATS, REC, REWARD, FDIST>), grid, dim3(3 * 2 * COVO_WAVE), 0, s, A, batch);
    else
        hipLaunchKernelGGL((rollout_pipe3_kernel<DISC1, ROLL, 2, 1, BATCHED, -1, 3, STATS, REC, REWARD, FDIST>), grid, dim3(3 * COVO_WAVE), 0, s, A, batch);
}

// the run-time switches of one (DISC1, BATCHED, REWARD, FDIST) family: rollover termination, position statistics (never
// batched), softmax records (A.records, all instances of a batch alike)
template <bool DISC1, bool BATCHED, int REWARD, int FDIST>
static void launch_pipe3_family(const RolloutArgs &A, const RolloutArgs *batch, int nb, int groups, bool stats, hipStream_t s)
{
    const bool rec = A.records != nullptr;
#define RP3_GO(ROLL, STATS, REC) launch_pipe3_groups<DISC1, ROLL, BATCHED, STATS, REC, REWARD, FDIST>(A, batch, nb, groups, s)
    if constexpr (!BATCHED) {
        if (stats) {
            if (A.rollover) { if (rec) RP3_GO(true, true, true); else RP3_GO(true, true, false); }
            else            { if (rec) RP3_GO(false, true, true); else RP3_GO(false, true, false); }
            return;
        }
    }
    if (A.rollover) { if (rec) RP3_GO(true, false, true); else RP3_GO(true, false, false); }
    else            { if (rec) RP3_GO(false, false, true); else RP3_GO(false, false, false); }
#undef RP3_GO
}

// the non-default reward / disturbance variants (rollout_var_r0.hip: REWARD 0 with FDIST 1, 2; rollout_var_r1.hip: REWARD 1);
// always the general discount path (acc = fma(discount^k, r, acc): bit-identical to acc + r at discount 1)
void launch_rollout_variant_r0(const RolloutArgs &A, const RolloutArgs *batch, int nb, bool batched, int groups, bool stats, hipStream_t s);
void launch_rollout_variant_r1(const RolloutArgs &A, const RolloutArgs *batch, int nb, bool batched, int groups, bool stats, hipStream_t s);
