// chol_probe.hip -- cycle budget of the single-workgroup 128x128 fp64 LDS Cholesky (chol_lds.hpp).
// Reports kernel time and clock64() (s_memtime) stamps: panel factor vs trailing update; checks against a host Cholesky.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CHOL_PROBE 1
#include "../../covo_mpc_amd/csrc/chol_lds.hpp"

constexpr int N = 128, LD = 144;

template <int VARIANT>
__global__ __launch_bounds__(512) void chol_k(const double *__restrict__ Ain, double *__restrict__ Lout, long long *stamps)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x;
    for (int e = tid; e < N * N; e += 512) sm[(e % N) * LD + e / N] = Ain[e];
    __syncthreads();
    const long long t0 = clock64();
    if (VARIANT == 0) chol_lds_fast(sm, N, LD, tid, 512);
    else chol128_lds_mfma<LD>(sm, tid);
    const long long t1 = clock64();
    for (int e = tid; e < N * N; e += 512) {
        const int r = e / N, c = e % N;
        Lout[e] = (c <= r) ? sm[c * LD + r] : 0.0;
    }
    if (tid == 0) {
        stamps[0] = t1 - t0;
        for (int i = 0; i < 24; ++i) stamps[1 + i] = chol_prof[i];
    }
}

template <int VARIANT>
void run(const char *name, const double *dA, double *dL, long long *dS, const std::vector<double> &ref)
{
    const size_t lds = (size_t)N * LD * sizeof(double);
    hipFuncSetAttribute(reinterpret_cast<const void *>(chol_k<VARIANT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(chol_k<VARIANT>, dim3(1), dim3(512), lds, 0, dA, dL, dS);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<double> L(N * N);
    long long st[32];
    hipMemcpy(L.data(), dL, N * N * 8, hipMemcpyDeviceToHost);
    hipMemcpy(st, dS, sizeof(st), hipMemcpyDeviceToHost);
    double err = 0, nrm = 0;
    for (int i = 0; i < N * N; ++i) { err = fmax(err, fabs(L[i] - ref[i])); nrm = fmax(nrm, fabs(ref[i])); }
    printf("%-10s kernel %7.2f us   chol %8lld ticks  [factor %lld  trsm %lld  update %lld  - %lld]  max err %.3e (rel %.3e)\n", name, best * 1e3, st[0],
           st[1], st[2], st[3], st[4], err, err / nrm);
    if (VARIANT) { printf("   update per panel:"); for (int q = 0; q < 8; ++q) printf(" %lld", st[9 + q]); printf("\n   factor per panel:"); for (int q = 0; q < 8; ++q) printf(" %lld", st[17 + q]); printf("\n"); }
}

int main()
{
    std::vector<double> A(N * N), G(N * N), ref(N * N, 0.0);
    srand(1);
    for (auto &g : G) g = (rand() / (double)RAND_MAX) - 0.5;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double s = (i == j) ? 0.05 : 0.0;
            for (int k = 0; k < N; ++k) s += G[i * N + k] * G[j * N + k] / N;
            A[i * N + j] = s;
        }
    std::vector<double> W = A;
    for (int j = 0; j < N; ++j) {
        double d = W[j * N + j];
        for (int k = 0; k < j; ++k) d -= ref[j * N + k] * ref[j * N + k];
        d = sqrt(d);
        ref[j * N + j] = d;
        for (int i = j + 1; i < N; ++i) {
            double s = W[i * N + j];
            for (int k = 0; k < j; ++k) s -= ref[i * N + k] * ref[j * N + k];
            ref[i * N + j] = s / d;
        }
    }
    double *dA, *dL;
    long long *dS;
    hipMalloc(&dA, N * N * 8);
    hipMalloc(&dL, N * N * 8);
    hipMalloc(&dS, 256);
    hipMemcpy(dA, A.data(), N * N * 8, hipMemcpyHostToDevice);
    run<0>("valu-p8", dA, dL, dS, ref);
    run<1>("mfma-p16", dA, dL, dS, ref);
    return 0;
}
