// rollout_lab.hip -- A/B bench of rollout kernel variants against the plain one-lane-per-sample kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -I covo_mpc_amd/csrc scripts/probe/rollout_lab.hip -o scripts/probe/rollout_lab
// Prints per variant: us per launch (back-to-back launches between two events), GB/s of algorithmic bytes (516 B/sample),
// and the largest relative cost difference to the plain kernel.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#define ROLLOUT_LAB_BASELINE
#include "../../covo_mpc_amd/csrc/rollout.hip"
#include "rollout_pipe4.hpp"  // the round-4 four-stage experiment (not faster; lives with the probes)

void covo_set_error(const char *fmt, ...) { (void)fmt; }
// what rollout.hip expects from the rest of the library (not linked into the lab)
int noise_gemm_groups_per_workgroup(int, int) { return 4; }
void launch_rollout_variant_r0(const RolloutArgs &, const RolloutArgs *, int, bool, int, bool, hipStream_t) {}
void launch_rollout_variant_r1(const RolloutArgs &, const RolloutArgs *, int, bool, int, bool, hipStream_t) {}

// ---- round 6: the TIMING BOUND of a split-horizon launch (VERDICT r05 Next 5).  Every 64 samples get SIX waves: two complete
// three-stage pipelines, each running 16 of the 32 steps on its own rings (the second reads stripes 16 .. 31), no join, no
// composition arithmetic -- i.e. exactly today's instruction streams and HBM bytes at twice the waves per SIMD.  A real
// split-horizon kernel adds the relative-to-absolute composition of the second half (+ ~35 VALU per step there) and a join on
// (q, v, p)_16: it cannot be faster than this.  Costs are garbage.
template <int CH, int GROUPS>
__global__ __launch_bounds__(6 * GROUPS * COVO_WAVE) void rollout_split_bound_kernel(const RolloutArgs A_, const size_t half_stride)
{
    __shared__ Rp3Lds<CH> lds_all[2 * GROUPS];
    __shared__ float lds_st[1][1][9];
    const int lane = threadIdx.x & (COVO_WAVE - 1);
    const int wave_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = wave_ / (3 * GROUPS), w = wave_ % (3 * GROUPS);
    const int role = w / GROUPS, gsub = w % GROUPS;
    RolloutArgs A = A_;
    if (half) A.a = A_.a + half_stride;  // stripes 16 .. 31
    const int group = blockIdx.x * GROUPS + gsub;
    float cost = 0.0f;
    bool valid = false;
    int n = 0;
    rp3_stages<true, false, CH, -1, false, false, 0, 0, false, decltype(lds_st), COVO_H / 2>(A, lds_all[half * GROUPS + gsub], lds_st, role, gsub,
                                                                                              group, lane, nullptr, cost, valid, n);
}

static float time_launches(const std::function<void()> &fn, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) fn();
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) fn();
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms * 1e3f / reps);
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return best;
}

int main(int argc, char **argv)
{
    const int T = 320;
    std::vector<int> Ns = {8192, 32768, 65536, 131072, 1048576};
    if (argc > 1) { Ns.clear(); for (int i = 1; i < argc; ++i) Ns.push_back(atoi(argv[i])); }
    covo_env_params prm;
    std::memset(&prm, 0, sizeof(prm));
    prm.max_thrust = 0.8f; prm.max_torque[0] = prm.max_torque[1] = 9e-3f; prm.max_torque[2] = 2e-3f;
    prm.max_omega[0] = prm.max_omega[1] = 10.f; prm.max_omega[2] = 3.f;
    prm.dt = 0.02f; prm.g = 9.81f; prm.m = 0.027f; prm.action_scale = 1.f; prm.alpha_bodyrate = 0.5f;
    prm.max_steps_in_episode = 300; prm.pos_limit = 3.0f;
    std::vector<float> st(COVO_STATE_FLOATS, 0.f), ptraj(T * 3), vtraj(T * 3);
    st[ST_POS + 0] = 0.11f; st[ST_POS + 1] = -0.07f; st[ST_POS + 2] = 0.03f;
    st[ST_VEL + 0] = 0.4f; st[ST_VEL + 1] = -0.2f; st[ST_VEL + 2] = 0.1f;
    st[ST_QUAT + 0] = 0.03f; st[ST_QUAT + 1] = -0.02f; st[ST_QUAT + 2] = 0.05f; st[ST_QUAT + 3] = 0.9979f;
    st[ST_OMEGA + 0] = 0.3f; st[ST_OMEGA + 1] = -0.1f; st[ST_OMEGA + 2] = 0.2f;
    st[ST_FDIST + 0] = 0.01f; st[ST_FDIST + 1] = -0.02f; st[ST_FDIST + 2] = 0.015f;
    const int time0 = 280;  // the horizon crosses max_steps: the done-freeze is exercised
    std::memcpy(&st[ST_TIME], &time0, 4);
    for (int i = 0; i < T; ++i) {
        ptraj[3 * i + 0] = 0.012f * i; ptraj[3 * i + 1] = -0.006f * i; ptraj[3 * i + 2] = 0.3f * sinf(0.05f * i);
        vtraj[3 * i + 0] = 0.6f; vtraj[3 * i + 1] = -0.3f; vtraj[3 * i + 2] = 0.75f * cosf(0.05f * i);
    }
    for (int j = 0; j < 3; ++j) { st[ST_POSTAR + j] = ptraj[3 * time0 + j] + 0.01f; st[ST_VELTAR + j] = vtraj[3 * time0 + j]; }
    float *dst, *dpt, *dvt;
    hipMalloc(&dst, st.size() * 4); hipMalloc(&dpt, ptraj.size() * 4); hipMalloc(&dvt, vtraj.size() * 4);
    hipMemcpy(dst, st.data(), st.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dpt, ptraj.data(), ptraj.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dvt, vtraj.data(), vtraj.size() * 4, hipMemcpyHostToDevice);
    const float fsh[3] = {0.02f, -0.01f, 0.03f};
    for (int N : Ns) {
        std::vector<float> a((size_t)COVO_H * N * 4);
        unsigned s = 12345u;
        for (auto &v : a) { s = s * 1664525u + 1013904223u; v = std::max(-1.f, std::min(1.f, ((float)(s >> 8) / 8388608.f - 1.f) * 1.2f)); }
        float *da, *dc, *dc2, *dg;
        hipMalloc(&da, a.size() * 4); hipMalloc(&dc, (size_t)N * 4); hipMalloc(&dc2, (size_t)N * 4);
        hipMalloc(&dg, (size_t)(N / 64 + 1) * 4);
        hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
        RolloutArgs A;
        fill_rollout_args(A, dst, dpt, dvt, T, prm, fsh, da, N, 1.0f, dc, dg, nullptr, nullptr, nullptr);
        RolloutArgs A2 = A;
        A2.cost = dc2;
        A.clip = A2.clip = 0;
        const int grid = (N + RO_BLOCK - 1) / RO_BLOCK;
        std::vector<float> ref(N), out(N);
        auto report = [&](const char *name, const std::function<void()> &fn, bool check) {
            const float us = time_launches(fn, 50);
            double worst = 0.0;
            if (check) {
                hipMemset(dc2, 0xff, (size_t)N * 4);
                fn();
                hipDeviceSynchronize();
                hipMemcpy(out.data(), dc2, (size_t)N * 4, hipMemcpyDeviceToHost);
                for (int i = 0; i < N; ++i) {
                    const double d = std::fabs((double)out[i] - ref[i]) / std::max(1.0, std::fabs((double)ref[i]));
                    worst = (d == d) ? std::max(worst, d) : 1e30;
                }
            }
            printf("N=%8d %-28s %8.2f us  %7.0f GB/s  %5.1f %%  maxrel %.2e\n", N, name, us, N * 516.0 / us / 1e3,
                   N * 516.0 / us / 1e3 / 80.0, worst);
            fflush(stdout);
        };
        hipLaunchKernelGGL((rollout_kernel<false, true, false, COVO_H>), dim3(grid), dim3(RO_BLOCK), 0, 0, A, nullptr);
        hipDeviceSynchronize();
        hipMemcpy(ref.data(), dc, (size_t)N * 4, hipMemcpyDeviceToHost);
        double mean = 0; for (float v : ref) mean += v; mean /= N;
        printf("N=%8d mean cost %.4f  cost[0..3] %.5f %.5f %.5f %.5f\n", N, mean, ref[0], ref[1], ref[2], ref[3]);
        report("plain PF=32", [&] { hipLaunchKernelGGL((rollout_kernel<false, true, false, COVO_H>), dim3(grid), dim3(RO_BLOCK), 0, 0, A2, nullptr); }, true);
        report("plain PF=8", [&] { hipLaunchKernelGGL((rollout_kernel<false, true, false, 8>), dim3(grid), dim3(RO_BLOCK), 0, 0, A2, nullptr); }, true);
#define PIPE3(CH, G)                                                                                                             \
        report("pipe3 CH=" #CH " GROUPS=" #G, [&] {                                                                               \
            hipLaunchKernelGGL((rollout_pipe3_kernel<true, false, CH, G>), dim3((N + 64 * G - 1) / (64 * G)), dim3(192 * G), 0, 0, A2, nullptr); }, true)
        PIPE3(1, 1); PIPE3(2, 1); PIPE3(1, 2); PIPE3(1, 4); PIPE3(2, 4);
#define PIPE4(CH, G)                                                                                                             \
        report("pipe4 CH=" #CH " GROUPS=" #G, [&] {                                                                               \
            hipLaunchKernelGGL((rollout_pipe4_kernel<true, false, CH, G>), dim3((N + 64 * G - 1) / (64 * G)), dim3((2 + CH) * 64 * G), 0, 0, A2, nullptr); }, true)
        PIPE4(2, 1); PIPE4(2, 2); PIPE4(2, 4); PIPE4(4, 1); PIPE4(4, 2); PIPE4(1, 4);
        report("pipe4 CH=2 GROUPS=4 +rollover", [&] { hipLaunchKernelGGL((rollout_pipe4_kernel<true, true, 2, 4>), dim3((N + 255) / 256), dim3(1024), 0, 0, A2, nullptr); }, false);
        report("pipe4 CH=2 GROUPS=4 discount", [&] { hipLaunchKernelGGL((rollout_pipe4_kernel<false, false, 2, 4>), dim3((N + 255) / 256), dim3(1024), 0, 0, A2, nullptr); }, true);
        report("pipe3 CH=1 GROUPS=4 +rollover", [&] { hipLaunchKernelGGL((rollout_pipe3_kernel<true, true, 1, 4>), dim3((N + 255) / 256), dim3(768), 0, 0, A2, nullptr); }, false);
        report("pipe3 CH=1 GROUPS=4 discount", [&] { hipLaunchKernelGGL((rollout_pipe3_kernel<false, false, 1, 4>), dim3((N + 255) / 256), dim3(768), 0, 0, A2, nullptr); }, true);
        report("pipe3 CH=2 no barriers (garbage)", [&] { hipLaunchKernelGGL((rollout_pipe3_kernel<true, false, 2, 1, false, -2>), dim3((N + 63) / 64), dim3(192), 0, 0, A2, nullptr); }, false);
#define SPLIT(CH, G)                                                                                                             \
        report("split-horizon BOUND CH=" #CH " GROUPS=" #G " (6 waves / 64 samples, garbage)", [&] {                               \
            hipLaunchKernelGGL((rollout_split_bound_kernel<CH, G>), dim3((N + 64 * G - 1) / (64 * G)), dim3(384 * G), 0, 0, A2, (size_t)16 * N); }, false)
        SPLIT(2, 1); SPLIT(2, 2); SPLIT(1, 2); SPLIT(4, 2);
#define STAGE(ROLE, WAVES)                                                                                                       \
        report("stage " #ROLE " alone, " #WAVES " waves/group", [&] {                                                              \
            hipLaunchKernelGGL((rollout_pipe3_kernel<true, false, 1, 1, false, ROLE, WAVES>), dim3((N + 63) / 64), dim3(64 * WAVES), 0, 0, A2, nullptr); }, false)
        if (getenv("LAB_STAGES")) { STAGE(0, 1); STAGE(0, 2); STAGE(0, 3); STAGE(0, 4); STAGE(1, 1); STAGE(1, 2); STAGE(1, 3); STAGE(1, 4); STAGE(2, 1); STAGE(2, 2); STAGE(2, 3); STAGE(2, 4); }
        hipFree(da); hipFree(dc); hipFree(dc2); hipFree(dg);
    }
    return 0;
}
