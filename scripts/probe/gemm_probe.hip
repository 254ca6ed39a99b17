// gemm_probe.hip -- phase timeline of the noise GEMM (covo_mpc_amd/csrc/noise_gemm.hip compiled with GEMM_PROBE):
// wave 0 of every workgroup stamps s_memrealtime at: kernel entry, L staged, after each k-group's MFMAs, after the last
// store.  Prints the event-timed duration and the median stamps (us from the earliest entry) for the in-kernel-Philox
// and the tile-ordered-epsilon variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -I covo_mpc_amd/csrc scripts/probe/gemm_probe.hip -o scripts/probe/gemm_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define GEMM_PROBE 1
#include "../../covo_mpc_amd/csrc/noise_gemm.hip"
void covo_set_error(const char *fmt, ...) { (void)fmt; }

__global__ void gen_kernel(EpsGenArgs G) { eps_tiles_generate(G, blockIdx.x * 4 + (threadIdx.x >> 6), gridDim.x * 4, threadIdx.x & 63); }

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 65536;
    std::vector<float> L(128 * 128, 0.f), mu(128, 0.f);
    for (int i = 0; i < 128; ++i)
        for (int k = 0; k <= i; ++k) L[i * 128 + k] = (i == k) ? 0.3f : 0.01f * (float)((i * 7 + k * 3) % 11 - 5);
    float *dL, *dmu, *da;
    float4 *deps;
    uint32_t *ddyn;
    unsigned long long *dprobe;
    hipMalloc(&dL, L.size() * 4); hipMalloc(&dmu, 512); hipMalloc(&da, (size_t)N * 512);
    hipMalloc(&deps, (size_t)((N + 31) / 32) * 16 * 64 * 16); hipMalloc(&ddyn, 48);
    hipMemcpy(dL, L.data(), L.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dmu, mu.data(), 512, hipMemcpyHostToDevice);
    uint32_t key[12] = {123u, 456u};
    hipMemcpy(ddyn, key, 48, hipMemcpyHostToDevice);
    const int nwg = 512;
    hipMalloc(&dprobe, (size_t)nwg * 64);
    hipMemcpyToSymbol(HIP_SYMBOL(g_ng_probe), &dprobe, sizeof(dprobe));
    EpsGenArgs G{deps, ddyn, 0, N};
    hipLaunchKernelGGL(gen_kernel, dim3(512), dim3(256), 0, 0, G);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; ++variant) {
        float best = 1e9f;
        for (int it = 0; it < 6; ++it) {
            hipEventRecord(e0);
            if (variant == 0) launch_noise_gemm(dL, dmu, nullptr, 123u, 456u, 0, N, da, 0);
            else launch_noise_gemm(dL, dmu, reinterpret_cast<const float *>(deps), 0, 0, 0, N, da, 0, nullptr, nullptr, 0, 1, true);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms * 1e3f);
        }
        const int ntiles = (N + 31) / 32, grid = std::min(512, (ntiles + 3) / 4);
        std::vector<unsigned long long> p((size_t)grid * 8);
        hipMemcpy(p.data(), dprobe, p.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < grid; ++w) t0 = std::min(t0, p[8 * w]);
        printf("N=%d %-14s %6.2f us (events) | median stamps (us):", N, variant == 0 ? "philox-in-GEMM" : "tiled epsilon", best);
        const char *lab[8] = {"entry", "L staged", "g0", "g1", "g2", "g3", "L loads landed", "L loads issued"};
        for (int i : {0, 7, 6, 1, 2, 3, 4, 5}) {
            std::vector<double> v;
            for (int w = 0; w < grid; ++w) v.push_back((double)(p[8 * w + i] - t0) * 0.01);
            std::sort(v.begin(), v.end());
            printf("  %s %.2f (max %.2f)", lab[i], v[v.size() / 2], v.back());
        }
        printf("\n");
    }
    return 0;
}
