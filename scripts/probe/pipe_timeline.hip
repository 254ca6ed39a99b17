// pipe_timeline.hip -- s_memrealtime stamps of the pipelined rollout's waves (chunk boundaries), N = 65 536.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define RP_TIMELINE 1
#include "../../covo_mpc_amd/csrc/rollout.hip"
void covo_set_error(const char *fmt, ...) { (void)fmt; }
#ifndef TL_CH
#define TL_CH 2
#endif
int main(int argc, char **argv)
{
    const int T = 320, N = argc > 1 ? atoi(argv[1]) : 65536;
    covo_env_params prm;
    std::memset(&prm, 0, sizeof(prm));
    prm.max_thrust = 0.8f; prm.max_torque[0] = prm.max_torque[1] = 9e-3f; prm.max_torque[2] = 2e-3f;
    prm.max_omega[0] = prm.max_omega[1] = 10.f; prm.max_omega[2] = 3.f;
    prm.dt = 0.02f; prm.g = 9.81f; prm.m = 0.027f; prm.action_scale = 1.f; prm.alpha_bodyrate = 0.5f;
    prm.max_steps_in_episode = 300; prm.pos_limit = 3.0f;
    std::vector<float> st(COVO_STATE_FLOATS, 0.f), traj(T * 3);
    st[ST_QUAT + 3] = 1.f;
    for (int i = 0; i < T * 3; ++i) traj[i] = 0.01f * (float)(i / 3) * ((i % 3) == 0 ? 1.f : -0.5f);
    float *dst, *dpt, *dvt;
    hipMalloc(&dst, st.size() * 4); hipMalloc(&dpt, traj.size() * 4); hipMalloc(&dvt, traj.size() * 4);
    hipMemcpy(dst, st.data(), st.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dpt, traj.data(), traj.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dvt, traj.data(), traj.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> a((size_t)COVO_H * N * 4);
    unsigned s = 12345u;
    for (auto &v : a) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 8388608.f - 1.f) * 0.6f; }
    float *da, *dc, *dg;
    hipMalloc(&da, a.size() * 4); hipMalloc(&dc, (size_t)N * 4); hipMalloc(&dg, (size_t)(N / 64 + 1) * 4);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    const int nwg = N / 64;
    unsigned long long *dtl;
    hipMalloc(&dtl, (size_t)nwg * 3 * 8 * 8);
    hipMemset(dtl, 0, (size_t)nwg * 3 * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_rp_tl), &dtl, sizeof(dtl));
    RolloutArgs A;
    fill_rollout_args(A, dst, dpt, dvt, T, prm, nullptr, da, N, 1.0f, dc, dg, nullptr, nullptr);
    for (int it = 0; it < 4; ++it) hipLaunchKernelGGL((rollout_pipe3_kernel<true, false, TL_CH, 1>), dim3(nwg), dim3(192), 0, 0, A, nullptr);
    hipDeviceSynchronize();
    std::vector<unsigned long long> tl((size_t)nwg * 3 * 8);
    hipMemcpy(tl.data(), dtl, tl.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (int g = 0; g < nwg; ++g) for (int w = 0; w < 3; ++w) t0 = std::min(t0, tl[((size_t)g * 3 + w) * 8]);
    auto us = [&](int g, int w, int i) { return (double)(tl[((size_t)g * 3 + w) * 8 + i] - t0) * 0.01; };
    printf("wg: wave A start/first-chunk/mid/end  (simd cu)   |  wave B ...\n");
    for (int g = 0; g < nwg; g += (g < 16 ? 1 : 37)) {
        printf("wg %4d:", g);
        for (int w = 0; w < 3; ++w) {
            const unsigned hw = (unsigned)tl[((size_t)g * 3 + w) * 8 + 4];
            printf("  %c %5.2f %5.2f %5.2f %5.2f (s%u c%2u e%u w%u)", "ATR"[w], us(g, w, 0), us(g, w, 1), us(g, w, 2), us(g, w, 3),
                   (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7, hw & 15);
        }
        printf("\n");
    }
    {
        std::vector<double> f;
        for (int g = 0; g < nwg; ++g) for (int w = 0; w < 3; ++w) f.push_back((double)tl[((size_t)g * 3 + w) * 8 + 5] / ((us(g, w, 3) - us(g, w, 1)) * 1e3));
        std::sort(f.begin(), f.end());
        printf("shader clock between first chunk and end: median %.2f GHz (min %.2f max %.2f)\n", f[f.size() / 2], f.front(), f.back());
    }
    for (int w = 0; w < 3; ++w) {
        printf("median wave %c:", "ATR"[w]);
        for (int i = 0; i < 4; ++i) {
            std::vector<double> v;
            for (int g = 0; g < nwg; ++g) v.push_back(us(g, w, i));
            std::sort(v.begin(), v.end());
            printf(" %5.2f (max %5.2f)", v[v.size() / 2], v.back());
        }
        printf("\n");
    }
    return 0;
}
