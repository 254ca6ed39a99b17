// rollout_probe.hip -- where and when the workgroups of the rollout kernels run (N = 65 536 and neighbours).
// Compiles covo_mpc_amd/csrc/rollout.hip with ROLLOUT_PROBE: every workgroup records {XCC, HW_ID, start, end}.
// Prints, per kernel variant: event-timed duration, workgroups per CU histogram, per-workgroup run time
// (min / median / max), and the span from the first start to the last end.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -I covo_mpc_amd/csrc scripts/probe/rollout_probe.hip -o scripts/probe/rollout_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
#define ROLLOUT_PROBE 1
#include "../../covo_mpc_amd/csrc/rollout.hip"

static thread_local char g_err[256];
void covo_set_error(const char *fmt, ...) { (void)fmt; }

static void report(const char *name, int nwg, unsigned long long *dprobe, float best_us)
{
    std::vector<unsigned long long> p(4 * (size_t)nwg);
    hipMemcpy(p.data(), dprobe, p.size() * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu;
    std::vector<double> dur;
    unsigned long long t_first = ~0ull, t_last = 0, s_last = 0;
    for (int i = 0; i < nwg; ++i) {
        const unsigned xcc = (unsigned)p[4 * i] & 0xf, hw = (unsigned)p[4 * i + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
        dur.push_back((double)(p[4 * i + 3] - p[4 * i + 2]) * 0.01);  // 100 MHz -> us
        t_first = std::min(t_first, p[4 * i + 2]);
        t_last = std::max(t_last, p[4 * i + 3]);
        s_last = std::max(s_last, p[4 * i + 2]);
    }
    std::sort(dur.begin(), dur.end());
    int hist[16] = {0};
    for (auto &kv : per_cu) hist[std::min(kv.second, 15)]++;
    printf("%-26s %6.2f us  | %4d WGs on %3zu CUs, WGs/CU histogram:", name, best_us, nwg, per_cu.size());
    for (int i = 1; i < 16; ++i)
        if (hist[i]) printf(" %dx%d", hist[i], i);
    printf(" | WG run time min/med/max %.2f/%.2f/%.2f us | first start -> last start %.2f, -> last end %.2f us\n", dur.front(),
           dur[dur.size() / 2], dur.back(), (double)(s_last - t_first) * 0.01, (double)(t_last - t_first) * 0.01);
}

int main(int argc, char **argv)
{
    const int T = 320;
    std::vector<int> Ns = {32768, 65536, 131072};
    if (argc > 1) { Ns.clear(); for (int i = 1; i < argc; ++i) Ns.push_back(atoi(argv[i])); }
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
    for (int N : Ns) {
        std::vector<float> a((size_t)COVO_H * N * 4);
        unsigned s = 12345u;
        for (auto &v : a) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 8388608.f - 1.f) * 0.6f; }
        float *da, *dc, *dg;
        hipMalloc(&da, a.size() * 4); hipMalloc(&dc, (size_t)N * 4); hipMalloc(&dg, (size_t)(N / 64 + 1) * 4);
        hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
        unsigned long long *dprobe;
        const int max_wg = N / 64 + 8;
        hipMalloc(&dprobe, (size_t)max_wg * 32);
        hipMemcpyToSymbol(HIP_SYMBOL(g_ro_probe), &dprobe, sizeof(dprobe));
        RolloutArgs A;
        fill_rollout_args(A, dst, dpt, dvt, T, prm, nullptr, da, N, 1.0f, dc, dg, nullptr, nullptr);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int variant = 0; variant < 3; ++variant) {
            const int grid = (N + RO_BLOCK - 1) / RO_BLOCK, g2 = (N + RS_PAIRS * 64 - 1) / (RS_PAIRS * 64);
            float best = 1e9f;
            for (int it = 0; it < 8; ++it) {
                hipEventRecord(e0);
                if (variant == 0) hipLaunchKernelGGL((rollout_kernel<false, true, false, COVO_H>), dim3(grid), dim3(RO_BLOCK), 0, 0, A, nullptr);
                else if (variant == 1) hipLaunchKernelGGL((rollout_kernel<false, true, false, 8>), dim3(grid), dim3(RO_BLOCK), 0, 0, A, nullptr);
                else hipLaunchKernelGGL((rollout_split_kernel<true, false>), dim3(g2), dim3(2 * RS_PAIRS * 64), 0, 0, A, nullptr);
                hipEventRecord(e1);
                hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = std::min(best, ms * 1e3f);
            }
            char name[64];
            snprintf(name, sizeof(name), "N=%d %s", N, variant == 0 ? "plain PF=32" : variant == 1 ? "plain PF=8" : "split");
            report(name, variant == 2 ? g2 : grid, dprobe, best);
        }
        hipFree(da); hipFree(dc); hipFree(dg); hipFree(dprobe);
    }
    return 0;
}
