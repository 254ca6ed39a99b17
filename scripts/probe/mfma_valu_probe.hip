// mfma_valu_probe.hip -- do fp32 MFMAs and fp32 VALU work overlap on ONE SIMD of gfx950?
// 1024-thread workgroups, one per CU: waves 0-3 (one per SIMD) issue v_mfma_f32_32x32x2_f32 back to back on one accumulator,
// waves 4-15 (three per SIMD) run independent v_fma_f32 chains.  Times: MFMA waves alone, VALU waves alone, both together.
// If the pipes overlapped the mixed run would take max(a, b); if fp32 MFMAs occupy the vector ALUs it takes a + b.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value scripts/probe/mfma_valu_probe.hip -o scripts/probe/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// MODE bit 1: MFMA waves work, bit 2: VALU waves work; BF16: the MFMA is v_mfma_f32_32x32x16_f16 instead (other datapath)
template <int MODE, bool F16>
__global__ __launch_bounds__(1024) void k(float *out, int n_mfma, int n_fma)
{
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (MODE & 1) {
            f32x16 acc;
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
            f16x8 ah, bh;
            for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)a; bh[e] = (_Float16)b; }
            for (int i = 0; i < n_mfma; ++i) {
                if (F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            for (int e = 0; e < 16; ++e) r += acc[e];
        }
    } else if (MODE & 2) {
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        const float c = 1.0001f, d = 0.5f;
        for (int i = 0; i < n_fma; ++i) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                         "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        }
        r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
    out[blockIdx.x * 1024 + threadIdx.x] = r;
}

template <int MODE, bool F16>
float run(float *out, int n_mfma, int n_fma)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, F16>), dim3(256), dim3(1024), 0, 0, out, n_mfma, n_fma);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    return best * 1e3f;
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 1024 * 4);
    const int n_mfma = 2000;  // x 64 cycles = 128k cycles per SIMD
    for (int n_fma : {2000, 4000, 8000}) {  // x 8 FMAs x 3 waves per SIMD
        const float a = run<1, false>(out, n_mfma, n_fma), b = run<2, false>(out, n_mfma, n_fma), ab = run<3, false>(out, n_mfma, n_fma);
        printf("fp32 32x32x2 MFMA x%d alone %8.1f us | 3 waves x %d x 8 v_fma alone %8.1f us | together %8.1f us (max %8.1f, sum %8.1f)\n", n_mfma, a,
               n_fma, b, ab, a > b ? a : b, a + b);
    }
    for (int n_fma : {2000, 4000, 8000}) {
        const float a = run<1, true>(out, 4000, n_fma), b = run<2, true>(out, 4000, n_fma), ab = run<3, true>(out, 4000, n_fma);
        printf("f16 32x32x16 MFMA x%d alone %8.1f us | 3 waves x %d x 8 v_fma alone %8.1f us | together %8.1f us (max %8.1f, sum %8.1f)\n", 4000, a,
               n_fma, b, ab, a > b ? a : b, a + b);
    }
    return 0;
}
