// barrier_probe.hip -- how cheap can a grid barrier between dependent 128^3 fp64 GEMM phases get on gfx950?
// One persistent launch, nw workgroups (spread over the XCDs), P phases C <- A.A * 1e-2 ping-pong, each tile computed by
// one workgroup (K split over its 4 waves); between phases a counter barrier.  Variants:
//   fence = 1: cached loads/stores + agent-scope release/acquire fences (buffer_wbl2 / buffer_inv: what sigma_ns.hip's
//              tail launch does today);  fence = 0: every data access is an agent-scope relaxed atomic (sc1: coherent
//              across the XCDs' L2s), the barrier is only s_waitcnt + the counter.
//   sleep: s_sleep argument between polls of the counter.
// Checks the result against the same chain run as separate launches.  Prints us per phase.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probe/barrier_probe.hip -o scripts/probe/barrier_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int SN = 128;

template <bool COH>
__device__ __forceinline__ double ld(const double *p)
{
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COH>
__device__ __forceinline__ void st(double *p, double v)
{
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

template <bool COH>
__device__ __forceinline__ void tile(const double *__restrict__ A, double *__restrict__ C, int t, double (*red)[4][64])
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lo = lane & 15, hi = lane >> 4;
    const int ti = t >> 3, tj = t & 7;
    double a[8], b[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int k = 32 * wv + 4 * kk + hi;
        a[kk] = ld<COH>(A + (size_t)k * SN + 16 * ti + lo);
        b[kk] = ld<COH>(A + (size_t)k * SN + 16 * tj + lo);
    }
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv][r][lane] = acc[r];
    __syncthreads();
    const double v = (red[0][wv][lane] + red[1][wv][lane]) + (red[2][wv][lane] + red[3][wv][lane]);
    st<COH>(C + (size_t)(16 * ti + hi + 4 * wv) * SN + 16 * tj + lo, v * 1e-2);
    __syncthreads();
}

template <bool FENCE, int SLEEP>
__device__ __forceinline__ bool grid_barrier(unsigned *ctr, unsigned target)
{
    __shared__ int ok;
    if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        int good = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(SLEEP);
            if (wall_clock64() - t0 > 20000000LL) { good = 0; break; }
        }
        ok = good;
    }
    __syncthreads();
    if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok != 0;
}

// round 4: no atomics -- every workgroup stores the phase number into its own flag word (sc1 store), waves 0.. poll all nw words
// with one sc1 load per 64 (xcd_chain_probe: 1.85 us per phase at 64 workgroups on one XCD against 2.74 with the counter)
template <int SLEEP>
__device__ __forceinline__ bool flag_barrier(unsigned *flags, unsigned phase, int nw)
{
    __shared__ int ok[4];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv * 64 < nw) {
        if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        int good = 1;
        for (;;) {
            const unsigned v = (wv * 64 + lane < nw) ? __hip_atomic_load(flags + wv * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : phase;
            if (__builtin_amdgcn_ballot_w64(v < phase) == 0) break;
            __builtin_amdgcn_s_sleep(SLEEP);
            if (wall_clock64() - t0 > 20000000LL) { good = 0; break; }
        }
        if (lane == 0) ok[wv] = good;
    } else if (lane == 0) ok[wv] = 1;
    __syncthreads();
    return (ok[0] & ok[1] & ok[2] & ok[3]) != 0;
}

template <bool FENCE, int SLEEP, bool FLAGS = false>
__global__ __launch_bounds__(256) void persist_k(double *b0, double *b1, unsigned *ctr, int phases)
{
    __shared__ double red[4][4][64];
    const int nw = gridDim.x;
    double *in = b0, *out = b1;
    for (int p = 0; p < phases; ++p) {
        for (int t = blockIdx.x; t < 64; t += nw) tile<!FENCE>(in, out, t, red);
        if (FLAGS) {
            if (!flag_barrier<SLEEP>(ctr + 64, (unsigned)(p + 1), nw)) return;
        } else if (!grid_barrier<FENCE, SLEEP>(ctr, (unsigned)(nw * (p + 1)))) return;
        double *x = in; in = out; out = x;
    }
}
__global__ __launch_bounds__(256) void one_k(const double *in, double *out)
{
    __shared__ double red[4][4][64];
    tile<false>(in, out, blockIdx.x, red);
}

template <bool FENCE, int SLEEP, bool FLAGS = false>
static void run(const char *name, int nw, double *dA, double *dB, unsigned *ctr, const std::vector<double> &h, const std::vector<double> &ref, int phases)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemcpy(dA, h.data(), SN * SN * 8, hipMemcpyHostToDevice);
        hipMemset(ctr, 0, 4 * 512);
        hipEventRecord(e0);
        hipLaunchKernelGGL((persist_k<FENCE, SLEEP, FLAGS>), dim3(nw), dim3(256), 0, 0, dA, dB, ctr, phases);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<double> out(SN * SN);
    hipMemcpy(out.data(), (phases & 1) ? dB : dA, SN * SN * 8, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < SN * SN; ++i) err = fmax(err, fabs(out[i] - ref[i]));
    printf("%-34s nw=%3d  %7.2f us/phase  (total %8.1f us, %d phases)  max |diff| vs separate launches %.1e\n", name, nw,
           best * 1e3f / phases, best * 1e3f, phases, err);
}

int main()
{
    double *dA, *dB;
    unsigned *ctr;
    hipMalloc(&dA, SN * SN * 8); hipMalloc(&dB, SN * SN * 8); hipMalloc(&ctr, 4 * 512);
    std::vector<double> h(SN * SN);
    for (int i = 0; i < SN * SN; ++i) h[i] = (i % 129 == 0) ? 9.0 : 0.3 * ((i * 37) % 11 - 5);
    const int phases = 40;
    hipMemcpy(dA, h.data(), SN * SN * 8, hipMemcpyHostToDevice);
    for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(one_k, dim3(64), dim3(256), 0, 0, (p & 1) ? dB : dA, (p & 1) ? dA : dB);
    hipDeviceSynchronize();
    std::vector<double> ref(SN * SN);
    hipMemcpy(ref.data(), (phases & 1) ? dB : dA, SN * SN * 8, hipMemcpyDeviceToHost);
    for (int nw : {16, 32, 36, 64, 128}) {
        run<true, 1>("fences, cached data, sleep 1", nw, dA, dB, ctr, h, ref, phases);
        run<true, 16>("fences, cached data, sleep 16", nw, dA, dB, ctr, h, ref, phases);
        run<false, 1>("no fences, sc1 data, sleep 1", nw, dA, dB, ctr, h, ref, phases);
        run<false, 4>("no fences, sc1 data, sleep 4", nw, dA, dB, ctr, h, ref, phases);
        run<false, 16>("no fences, sc1 data, sleep 16", nw, dA, dB, ctr, h, ref, phases);
        run<false, 0, true>("sc1 data, flag words, sleep 0", nw, dA, dB, ctr, h, ref, phases);
        run<false, 1, true>("sc1 data, flag words, sleep 1", nw, dA, dB, ctr, h, ref, phases);
    }
    printf("final: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    return 0;
}
