// issue_probe.hip -- instruction-throughput probe for gfx950 (build: hipcc --offload-arch=gfx950 -O3).
// Prints shader cycles per wave-instruction (s_memtime) and wall-clock TFLOP/s for:
//   v_fma_f32, v_pk_fma_f32, v_mfma_f32_32x32x2_f32, v_mfma_f32_16x16x4_f32, v_fma_f64
// at 1, 2 and 4 waves per SIMD.  Used to size the rollout (VALU) and noise-GEMM (MFMA) kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../covo_mpc_amd/csrc/quad_model.hpp"
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int ITERS = 2048;

template <int KIND>
__global__ __launch_bounds__(256) void probe(float *out, unsigned long long *cyc, float seed)
{
    const int tid = threadIdx.x;
    float x = seed + tid * 1e-6f;
    unsigned long long t0 = 0, t1 = 0;
    if (KIND == 0) {  // 8 independent v_fma_f32 chains
        float a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < ITERS; ++i) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(0.999f), "v"(0.001f));
        }
        t1 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * 256 + tid] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else if (KIND == 1) {  // 8 independent v_pk_fma_f32 chains
        f32x2 a0 = {x, x}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
        f32x2 m = {0.999f, 0.999f}, c = {0.001f, 0.001f};
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < ITERS; ++i) {
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        }
        t1 = __builtin_amdgcn_s_memtime();
        f32x2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        out[blockIdx.x * 256 + tid] = s[0] + s[1];
    } else if (KIND == 2) {  // 4 accumulators of v_mfma_f32_32x32x2_f32
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < ITERS / 4; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x + 1, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x + 2, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x + 3, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x + 4, c3, 0, 0, 0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * 256 + tid] = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (KIND == 3) {  // 8 accumulators of v_mfma_f32_16x16x4_f32
        f32x4 c[8] = {};
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < ITERS / 4; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x + u, c[u], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
        for (int u = 0; u < 8; ++u) s += c[u][0];
        out[blockIdx.x * 256 + tid] = s;
    } else {  // 8 independent v_fma_f64 chains
        double a[8];
        for (int u = 0; u < 8; ++u) a[u] = x + u;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < ITERS; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[u]) : "v"(0.999), "v"(0.001));
        }
        t1 = __builtin_amdgcn_s_memtime();
        double s = 0;
        for (int u = 0; u < 8; ++u) s += a[u];
        out[blockIdx.x * 256 + tid] = (float)s;
    }
    if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// the real rollout arithmetic (quad_model.hpp), registers only: UNROLL = fully unrolled 32 steps
// (30 KB of straight-line code, like rollout_kernel) or a rolled loop (1 KB body)
template <bool UNROLL, int PART>
__global__ __launch_bounds__(256) void model_probe(float *out, unsigned long long *cyc, float seed)
{
    const int tid = threadIdx.x;
    qm::State<float> s;
    s.px = seed * 0.1f + tid * 1e-4f; s.py = 0.2f; s.pz = -0.1f; s.vx = 0.3f; s.vy = -0.2f; s.vz = 0.1f;
    s.qx = 0.05f; s.qy = -0.03f; s.qz = 0.02f; s.qw = 0.99f; s.ox = 0.3f; s.oy = -0.2f; s.oz = 0.1f;
    qm::Consts<float> c;
    c.thrust_half = 0.4f; c.komega[0] = 10.f; c.komega[1] = 10.f; c.komega[2] = 3.f; c.dt = 0.02f; c.half_dt = 0.01f;
    c.neg_g = -9.81f; c.inv_m = 37.037f; c.alpha = 0.5f; c.one_m_alpha = 0.5f; c.pos_limit = 3.0f;
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (UNROLL) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (PART != 1) acc += qm::reward<float, float>(s, 0.1f * k, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f);
            if (PART != 2) qm::dyn_step<float, float>(s, -0.33f + 0.001f * k, 0.1f, -0.1f, 0.05f, c, 0.01f, 0.02f, 0.03f);
        }
    } else {
#pragma unroll 1
        for (int k = 0; k < 32; ++k) {
            if (PART != 1) acc += qm::reward<float, float>(s, 0.1f * k, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f);
            if (PART != 2) qm::dyn_step<float, float>(s, -0.33f + 0.001f * k, 0.1f, -0.1f, 0.05f, c, 0.01f, 0.02f, 0.03f);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + tid] = acc + s.px + s.qw + s.ox;
    if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <bool UNROLL, int PART>
void run_model(const char *name, float *out, unsigned long long *cyc)
{
    for (int wps : {1, 2, 4, 8}) {
        const int grid = 256 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL((model_probe<UNROLL, PART>), dim3(grid), dim3(256), 0, 0, out, cyc, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((model_probe<UNROLL, PART>), dim3(grid), dim3(256), 0, 0, out, cyc, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c;
        hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        printf("%-34s waves/SIMD=%d  memtime-ticks(wave0)=%8llu  wall=%8.2f us\n", name, wps, c, ms * 1e3);
    }
}

template <int KIND>
void run(const char *name, double insts_per_wave, double flops_per_inst, float *out, unsigned long long *cyc)
{
    for (int wps : {1, 2, 4}) {
        const int grid = 256 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c;
        hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        const double waves = grid * 4.0;
        printf("%-28s waves/SIMD=%d  memtime-ticks/inst=%7.2f  wall=%8.2f us  %8.2f TFLOP/s\n", name, wps,
               (double)c / insts_per_wave, ms * 1e3, waves * insts_per_wave * flops_per_inst / (ms * 1e-3) / 1e12);
    }
}

int main()
{
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipMalloc(&cyc, sizeof(*cyc));
    run<0>("v_fma_f32 x8 chains", ITERS * 8.0, 64 * 2.0, out, cyc);
    run<1>("v_pk_fma_f32 x8 chains", ITERS * 8.0, 64 * 4.0, out, cyc);
    run<2>("v_mfma_f32_32x32x2_f32 x4", ITERS * 2.0, 32.0 * 32 * 2 * 2, out, cyc);
    run<3>("v_mfma_f32_16x16x4_f32 x8", ITERS * 2.0, 16.0 * 16 * 4 * 2, out, cyc);
    run<4>("v_fma_f64 x8 chains", ITERS * 8.0, 64 * 2.0, out, cyc);
    run_model<true, 0>("model 32 steps unrolled (rew+dyn)", out, cyc);
    run_model<false, 0>("model 32 steps rolled   (rew+dyn)", out, cyc);
    run_model<true, 1>("model unrolled, dynamics only", out, cyc);
    run_model<true, 2>("model unrolled, reward only", out, cyc);
    return 0;
}
