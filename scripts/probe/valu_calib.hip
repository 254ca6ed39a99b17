// valu_calib.hip -- what the SQ counters read for a VALU pipe whose load is KNOWN.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probe/valu_calib.hip -o scripts/probe/valu_calib
//   rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \
//       --output-format csv -d gpurun_out/valu_calib -o calib -- scripts/probe/valu_calib
// Every kernel puts W waves on each of the chip's 1024 SIMDs (256 CUs x W workgroups of 256 threads) and has every wave
// issue ITER x 32 v_fma_f32: KIND 0 one dependent chain, KIND 1 eight independent chains, KIND 2 the rollout's mix (7 fma +
// one v_rsq_f32).  The kernel name carries (KIND, W), so the per-dispatch counter rows can be read per configuration;
// scripts/valu_calib_summary.py turns them into cycles per instruction and the counter-derived "utilisation" figures.
// The binary also prints its own event-timed wall clock per launch (un-profiled runs: the reference for the rates).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(X) X X X X
#define REP8(X) REP4(X) REP4(X)
#define REP16(X) REP8(X) REP8(X)
#define REP32(X) REP16(X) REP16(X)

constexpr int ITER = 512;

template <int KIND, int W>
__global__ __launch_bounds__(256) void calib(float *out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float c = 1.0001f, d = 0.5f;
    for (int it = 0; it < ITER; ++it) {
        if (KIND == 0) {
            REP32(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(c), "v"(d));)
        } else if (KIND == 1) {
            REP4(asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                              "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else {
            REP4(asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                              "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_rsq_f32 %7, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND, int W>
static void run(float *out, const char *name)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((calib<KIND, W>), dim3(256 * W), dim3(256), 0, 0, out, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double n = (double)ITER * 32 * W;  // VALU wave-instructions per SIMD
    printf("%-24s W=%d  wall %8.2f us  %6.3f ns per wave-instruction per SIMD\n", name, W, best * 1e3, best * 1e6 / n);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main()
{
    float *out;
    hipMalloc(&out, (size_t)256 * 8 * 256 * 4);
#define ALLW(KIND, NAME) run<KIND, 1>(out, NAME); run<KIND, 2>(out, NAME); run<KIND, 3>(out, NAME); run<KIND, 4>(out, NAME); run<KIND, 6>(out, NAME); run<KIND, 8>(out, NAME)
    ALLW(0, "dependent v_fma_f32");
    ALLW(1, "8 independent v_fma_f32");
    ALLW(2, "7 fma + 1 rsq");
    hipFree(out);
    return 0;
}
