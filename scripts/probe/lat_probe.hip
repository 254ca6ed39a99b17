// lat_probe.hip -- dependent-chain latencies on gfx950 (one wave, cycles from s_memtime/clock64):
// fp64 FMA, v_rsq_f64, v_mfma_f64_16x16x4_f64 (same accumulator), MFMA -> v_readlane -> VALU -> MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int R = 256;
#define USE(v) asm volatile("" ::"v"(v))
__device__ __forceinline__ long long tick() { __builtin_amdgcn_sched_barrier(0); long long t = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); return t; }

__global__ void k(double *out, long long *cyc, double seed)
{
    const int lane = threadIdx.x;
    double x = seed + lane * 1e-3;
    long long t0, t1;
    // 1. dependent v_fma_f64
    t0 = tick();
#pragma unroll
    for (int i = 0; i < R; ++i) x = fma(x, 0.999, 1e-3);
    USE(x);
    t1 = tick();
    cyc[0] = t1 - t0;
    // 2. dependent v_rsq_f64
    double y = x + 2.0;
    t0 = tick();
#pragma unroll
    for (int i = 0; i < R; ++i) y = __builtin_amdgcn_rsq(y);
    USE(y);
    t1 = tick();
    cyc[1] = t1 - t0;
    // 3. dependent MFMA f64 16x16x4 on one accumulator
    f64x4 acc = {x, y, x, y};
    t0 = tick();
#pragma unroll
    for (int i = 0; i < R; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
    USE(acc[0]);
    t1 = tick();
    cyc[2] = t1 - t0;
    // 4. two independent accumulators alternating
    f64x4 acc2 = {y, x, y, x};
    t0 = tick();
#pragma unroll
    for (int i = 0; i < R; ++i) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, acc2, 0, 0, 0);
    }
    USE(acc[0]); USE(acc2[0]);
    t1 = tick();
    cyc[3] = t1 - t0;
    // 5. MFMA -> readlane -> 1 fp64 mul -> MFMA operand
    double a = x;
    t0 = tick();
#pragma unroll
    for (int i = 0; i < R; ++i) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, y, acc, 0, 0, 0);
        const int lo_ = __builtin_amdgcn_readlane(__double2loint(acc[0]), 5);
        const int hi_ = __builtin_amdgcn_readlane(__double2hiint(acc[0]), 5);
        a = __hiloint2double(hi_, lo_) * 1e-3;
    }
    USE(a);
    t1 = tick();
    cyc[4] = t1 - t0;
    // 6. fp32 dependent fma
    float f = (float)x;
    t0 = tick();
#pragma unroll
    for (int i = 0; i < R; ++i) f = fmaf(f, 0.999f, 1e-3f);
    USE(f);
    t1 = tick();
    cyc[5] = t1 - t0;
    // 7. independent fp64 fma x4 streams
    double z0 = x, z1 = y, z2 = x + 1, z3 = y + 1;
    t0 = tick();
#pragma unroll
    for (int i = 0; i < R; ++i) {
        z0 = fma(z0, 0.999, 1e-3);
        z1 = fma(z1, 0.999, 1e-3);
        z2 = fma(z2, 0.999, 1e-3);
        z3 = fma(z3, 0.999, 1e-3);
    }
    USE(z0); USE(z1); USE(z2); USE(z3);
    t1 = tick();
    cyc[6] = t1 - t0;
    // 8. MFMA f32 16x16x4 dependent
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 c32 = {f, f, f, f};
    t0 = tick();
#pragma unroll
    for (int i = 0; i < R; ++i) c32 = __builtin_amdgcn_mfma_f32_16x16x4f32(f, 1.0f, c32, 0, 0, 0);
    USE(c32[0]);
    t1 = tick();
    cyc[7] = t1 - t0;
    out[lane] = x + y + acc[0] + acc2[1] + a + f + z0 + z1 + z2 + z3 + c32[0];
}

int main()
{
    double *o;
    long long *c;
    hipMalloc(&o, 64 * 8);
    hipMalloc(&c, 64);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, 1.0);
    long long h[8];
    hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
    const char *n[8] = {"dependent v_fma_f64", "dependent v_rsq_f64", "dependent mfma_f64_16x16x4", "2 independent mfma_f64 (per pair)",
                        "mfma -> readlane x2 -> v_mul_f64 -> mfma", "dependent v_fma_f32", "4 independent v_fma_f64 (per 4)", "dependent mfma_f32_16x16x4"};
    for (int i = 0; i < 8; ++i) printf("%-45s %7.1f ticks/iter\n", n[i], (double)h[i] / R);
    return 0;
}
