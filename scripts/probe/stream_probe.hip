// stream_probe.hip -- what HBM/L2 bandwidth does the rollout's action-read pattern get on gfx950?
// One lane = one sample reads 32 float4 (512 B) and writes 4 B.  Layouts:
//   stripe : a[k][n]            (stride N*16 B between a lane's consecutive loads)
//   tiled  : a[n/64][k][n%64]   (each wave reads one contiguous 32 KiB block)
// All 32 loads are issued up front (like rollout_kernel PF=32) or in a ring of 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <bool TILED, int PF>
__global__ __launch_bounds__(256) void stream_k(const float4 *__restrict__ a, float *__restrict__ out, int N)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float4 *p = TILED ? a + (size_t)(n >> 6) * 32 * 64 + (n & 63) : a + n;
    const size_t stride = TILED ? 64 : (size_t)N;
    float4 r[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) r[i] = p[(size_t)i * stride];
    __builtin_amdgcn_sched_barrier(0);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const float4 v = r[k % PF];
        if (PF < 32 && k + PF < 32) r[k % PF] = p[(size_t)(k + PF) * stride];
        acc += v.x + v.y * 2.f + v.z * 3.f + v.w * 4.f;
    }
    out[n] = acc;
}

template <bool TILED, int PF>
void run(const char *name, const float4 *a, float *out, int N, int flush_mb, float *flush)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    for (int it = 0; it < 6; ++it) {
        if (flush_mb) hipMemsetAsync(flush, it, (size_t)flush_mb << 20, 0);  // evict a from L2 / Infinity Cache
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream_k<TILED, PF>), dim3((N + 255) / 256), dim3(256), 0, 0, a, out, N);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it > 0) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("%-22s N=%8d flush=%4dMB  best %8.2f us (%6.0f GB/s)  avg %8.2f us\n", name, N, flush_mb, best * 1e3,
           N * 516.0 / (best * 1e-3) / 1e9, sum / 5 * 1e3);
}

int main()
{
    const int NMAX = 1 << 20;
    float4 *a;
    float *out, *flush;
    hipMalloc(&a, (size_t)NMAX * 32 * sizeof(float4));
    hipMalloc(&out, NMAX * sizeof(float));
    hipMalloc(&flush, (size_t)1024 << 20);
    hipMemset(a, 0, (size_t)NMAX * 32 * sizeof(float4));
    for (int flush_mb : {0, 1024})
        for (int N : {65536, 131072, 262144, 1048576}) {
            run<false, 32>("stripe PF=32", a, out, N, flush_mb, flush);
            run<false, 8>("stripe PF=8", a, out, N, flush_mb, flush);
            run<true, 32>("tiled  PF=32", a, out, N, flush_mb, flush);
            run<true, 8>("tiled  PF=8", a, out, N, flush_mb, flush);
        }
    return 0;
}
