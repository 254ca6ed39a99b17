// valu_probe.hip -- issue cost of the rollout's instruction kinds on gfx950, alone and sharing a SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value scripts/probe/valu_probe.hip -o scripts/probe/valu_probe
// Every kernel runs ITER iterations of a small unrolled body (fits the instruction cache) on W waves per SIMD of all
// 256 CUs and reports shader-clock ticks (s_memtime) per instruction per wave and per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define REP4(X) X X X X
#define REP8(X) REP4(X) REP4(X)
#define REP16(X) REP8(X) REP8(X)
#define REP32(X) REP16(X) REP16(X)

constexpr int ITER = 256;

template <int KIND, bool HALF = false>
__global__ __launch_bounds__(256) void probe(float *out, unsigned long long *ticks, float seed)
{
    if (HALF && (threadIdx.x & 63) >= 32) return;  // upper half of every wave masked off
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float c = 1.0001f, d = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pc = {c, c}, pd = {d, d};
    const unsigned long long r0 = wall_clock64();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
        if (KIND == 0) {  // dependent fma chain, 32 per iter
            REP32(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(c), "v"(d));)
        } else if (KIND == 1) {  // 8 independent fma chains, 32 per iter
            REP4(asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                              "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (KIND == 2) {  // dependent rsq chain
            REP32(asm volatile("v_rsq_f32 %0, %0" : "+v"(a0));)
        } else if (KIND == 3) {  // 8 independent rsq
            REP4(asm volatile("v_rsq_f32 %0, %0\n\tv_rsq_f32 %1, %1\n\tv_rsq_f32 %2, %2\n\tv_rsq_f32 %3, %3\n\t"
                              "v_rsq_f32 %4, %4\n\tv_rsq_f32 %5, %5\n\tv_rsq_f32 %6, %6\n\tv_rsq_f32 %7, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 4) {  // 7 independent fma + 1 rsq, x4 (our mix is ~1 transcendental per 21)
            REP4(asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                              "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_rsq_f32 %7, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (KIND == 5) {  // dependent pairs: fma -> fma (2 chains of 16)
            REP16(asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(c), "v"(d));)
        } else if (KIND == 6) {  // v_readlane + dependent use
            REP16(asm volatile("v_readlane_b32 s20, %1, 3\n\tv_fma_f32 %0, %0, s20, %2" : "+v"(a0) : "v"(a1), "v"(d) : "s20");)
        } else if (KIND == 7) {  // v_pk_fma_f32, 4 independent chains (register pairs), 32 per iter
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc), "v"(pd));)
        } else if (KIND == 14) {  // v_pk_mul_f32 / v_pk_add_f32 alternating, 4 chains
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %5"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc), "v"(pd));)
        } else if (KIND == 15) {  // dependent v_pk_fma_f32
            REP32(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pc), "v"(pd));)
        } else if (KIND == 8) {  // v_sqrt dependent
            REP32(asm volatile("v_sqrt_f32 %0, %0" : "+v"(a0));)
        } else if (KIND == 9) {  // v_log dependent
            REP32(asm volatile("v_log_f32 %0, %0" : "+v"(a0));)
        } else if (KIND == 10) {  // v_rcp dependent
            REP32(asm volatile("v_rcp_f32 %0, %0" : "+v"(a0));)
        } else if (KIND == 11) {  // cmp + cndmask pairs (vcc dependency)
            REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a0) : "v"(a1), "v"(a2) : "vcc");)
        } else if (KIND == 12) {  // 4 independent chains: mul, fma alternating (e32 encodings)
            REP8(asm volatile("v_mul_f32 %0, %0, %4\n\tv_fmac_f32 %1, %5, %4\n\tv_mul_f32 %2, %2, %4\n\tv_fmac_f32 %3, %5, %4"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));)
        } else if (KIND == 13) {  // 3 independent fma + 1 rsq, x8
            REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_rsq_f32 %3, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if ((threadIdx.x & 63) == 0) { ticks[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0; ticks[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0; }
}

template <int KIND, bool HALF = false>
void run(const char *name, int per_iter)
{
    float *out;
    unsigned long long *ticks;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMalloc(&ticks, 256 * 8 * 4 * 8 * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int W : {1, 2, 4, 8}) {
        const int grid = 256 * W;
        hipLaunchKernelGGL((probe<KIND, HALF>), dim3(grid), dim3(256), 0, 0, out, ticks, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<KIND, HALF>), dim3(grid), dim3(256), 0, 0, out, ticks, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> tt(grid * 4 * 2), t(grid * 4), r(grid * 4);
        hipMemcpy(tt.data(), ticks, tt.size() * 8, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < t.size(); ++i) { t[i] = tt[2 * i]; r[i] = tt[2 * i + 1]; }
        std::sort(t.begin(), t.end());
        std::sort(r.begin(), r.end());
        const double med = (double)t[t.size() / 2], rmed = (double)r[r.size() / 2] * 10.0;  // ns (100 MHz)
        const double n = (double)ITER * per_iter;
        printf("%-36s W=%d  ticks/inst/wave %6.2f  ns/inst/wave %6.2f  ns/inst/SIMD %5.2f  clk %4.2f GHz  wall %7.2f us\n", name, W,
               med / n, rmed / n, rmed / n / W, med / rmed, ms * 1e3);
    }
    hipFree(out);
    hipFree(ticks);
}

int main()
{
    run<1, true>("8 indep v_fma_f32, 32 lanes active", 32);
    run<5, true>("2 chains v_fma_f32, 32 lanes active", 32);
    run<4, true>("7 fma + 1 rsq, 32 lanes active", 32);
    run<0>("dependent v_fma_f32", 32);
    run<7>("4 chains v_pk_fma_f32", 32);
    run<14>("4 chains v_pk_mul/v_pk_add", 32);
    run<15>("dependent v_pk_fma_f32", 32);
    run<1>("8 independent v_fma_f32", 32);
    run<5>("2 chains v_fma_f32", 32);
    run<12>("4 chains v_mul/v_fmac e32", 32);
    run<2>("dependent v_rsq_f32", 32);
    run<3>("8 independent v_rsq_f32", 32);
    run<8>("dependent v_sqrt_f32", 32);
    run<9>("dependent v_log_f32", 32);
    run<10>("dependent v_rcp_f32", 32);
    run<4>("7 fma + 1 rsq (independent)", 32);
    run<13>("3 fma + 1 rsq (independent)", 32);
    run<6>("v_readlane -> v_fma (sgpr)", 32);
    run<11>("v_cmp -> v_cndmask (vcc)", 32);
    return 0;
}
