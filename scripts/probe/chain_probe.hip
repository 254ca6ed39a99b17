// chain_probe.hip -- what does one link of a DEPENDENT chain of tiny fp64 GEMM launches cost on gfx950,
// and what would the same chain cost inside one persistent launch confined to one XCD?
// Feeds the design of the eigh-free Sigma pipeline (sigma_ns.hip): ~70 dependent 128x128x128 fp64 GEMMs.
//   A  empty kernel <<<1,64>>>  chain (eager, then hipGraph)
//   B  empty kernel <<<64,256>>> chain
//   C  gemm1: one wave per 16x16 tile, K = 128 (what sigma_ns.hip does today), 16 WG x 256
//   D  gemm4: one 256-thread WG per tile, K split over the 4 waves, LDS reduce, 64 WG x 256
//   E  persistent: 32 WGs (blockIdx % 8 == 0 of a 256-WG grid -> one XCD if dispatch is round-robin) or
//      32 consecutive WGs (spread over the XCDs); P phases of gemm-like work separated by a counter barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int SN = 128;

__global__ void empty_k(int *p) { if (p && threadIdx.x == 9999) *p = 1; }

__device__ __forceinline__ f64x4 tile_mm_k(const double *__restrict__ A, const double *__restrict__ B, int ti, int tj, int lane,
                                           int k0, int nk)
{
    const int lo = lane & 15, hi = lane >> 4;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    double a[32], b[32];
#pragma unroll
    for (int kk = 0; kk < 32; ++kk)
        if (kk < nk) {
            a[kk] = A[(size_t)(k0 + 4 * kk + hi) * SN + 16 * ti + lo];
            b[kk] = B[(size_t)(k0 + 4 * kk + hi) * SN + 16 * tj + lo];
        }
#pragma unroll
    for (int kk = 0; kk < 32; ++kk)
        if (kk < nk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc, 0, 0, 0);
    return acc;
}

__global__ __launch_bounds__(256) void gemm1_k(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ C)
{
    const int lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ti = w >> 3, tj = w & 7;
    const f64x4 acc = tile_mm_k(A, B, ti, tj, lane, 0, 32);
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(size_t)(16 * ti + (lane >> 4) + 4 * r) * SN + 16 * tj + (lane & 15)] = acc[r] * 1e-2;
}

__global__ __launch_bounds__(256) void gemm4_k(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ C)
{
    __shared__ double red[3][4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, w = blockIdx.x;
    const int ti = w >> 3, tj = w & 7;
    f64x4 acc = tile_mm_k(A, B, ti, tj, lane, 32 * wv, 8);
    if (wv > 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv - 1][r][lane] = acc[r];
    __syncthreads();
    if (wv == 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double v = (acc[r] + red[0][r][lane]) + (red[1][r][lane] + red[2][r][lane]);
            C[(size_t)(16 * ti + (lane >> 4) + 4 * r) * SN + 16 * tj + (lane & 15)] = v * 1e-2;
        }
}

// gemm4 confined to one XCD: grid of 8 x 64 workgroups, only blockIdx % 8 == 0 work (round-robin dispatch puts them
// on the same XCD, whose L2 then serves every operand of the chain)
__global__ __launch_bounds__(256) void gemm4x_k(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ C)
{
    __shared__ double red[3][4][64];
    if (blockIdx.x & 7) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, w = blockIdx.x >> 3;
    const int ti = w >> 3, tj = w & 7;
    f64x4 acc = tile_mm_k(A, B, ti, tj, lane, 32 * wv, 8);
    if (wv > 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv - 1][r][lane] = acc[r];
    __syncthreads();
    if (wv == 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double v = (acc[r] + red[0][r][lane]) + (red[1][r][lane] + red[2][r][lane]);
            C[(size_t)(16 * ti + (lane >> 4) + 4 * r) * SN + 16 * tj + (lane & 15)] = v * 1e-2;
        }
}

__device__ __forceinline__ bool grid_barrier(unsigned *ctr, unsigned target)
{
    __shared__ int ok;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        int good = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 20000000LL) { good = 0; break; }  // 0.2 s at 100 MHz: bail out, never hang the box
        }
        ok = good;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok != 0;
}

// persistent: `nw` worker WGs; each phase = 128 wave-jobs (64 tiles x K-split 2, summed through LDS by the wave pair)
__global__ __launch_bounds__(256) void persist_k(double *__restrict__ buf0, double *__restrict__ buf1, unsigned *ctr, int phases,
                                                 int stride, int nw, int *xcc_out)
{
    __shared__ double red[2][4][64];
    if (blockIdx.x % stride != 0) return;
    const int wg = blockIdx.x / stride;
    if (wg >= nw) return;
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc_out[wg] = (int)(x & 0xf);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int jobs_per_wg = 128 / nw;  // nw = 32 -> 4 jobs = 1 per wave
    double *in = buf0, *out = buf1;
    for (int p = 0; p < phases; ++p) {
        for (int j = wv; j < jobs_per_wg; j += 4) {
            const int job = wg * jobs_per_wg + j;  // tile = job/2, khalf = job&1; pair = waves (2m, 2m+1)
            const int tile = job >> 1, kh = job & 1;
            const int ti = tile >> 3, tj = tile & 7;
            f64x4 acc = tile_mm_k(in, in, ti, tj, lane, 64 * kh, 16);
            if (kh)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wv >> 1][r][lane] = acc[r];
            __syncthreads();
            if (!kh)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[(size_t)(16 * ti + (lane >> 4) + 4 * r) * SN + 16 * tj + (lane & 15)] = (acc[r] + red[wv >> 1][r][lane]) * 1e-2;
        }
        if (!grid_barrier(ctr, (unsigned)(nw * (p + 1)))) return;
        double *t = in; in = out; out = t;
    }
}

static float time_chain(hipStream_t s, int n, void (*launch)(hipStream_t, int))
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, s);
        for (int i = 0; i < n; ++i) launch(s, i);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3f / n;
}

static float time_graph(hipStream_t s, int n, void (*launch)(hipStream_t, int))
{
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n; ++i) launch(s, i);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, s);
        hipGraphLaunch(ge, s);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3f / n;
}

static double *gA, *gB;
static void l_empty1(hipStream_t s, int) { hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s, nullptr); }
static void l_empty64(hipStream_t s, int) { hipLaunchKernelGGL(empty_k, dim3(64), dim3(256), 0, s, nullptr); }
static void l_gemm1(hipStream_t s, int i) { hipLaunchKernelGGL(gemm1_k, dim3(16), dim3(256), 0, s, (i & 1) ? gB : gA, (i & 1) ? gB : gA, (i & 1) ? gA : gB); }
static void l_gemm4x(hipStream_t s, int i) { hipLaunchKernelGGL(gemm4x_k, dim3(512), dim3(256), 0, s, (i & 1) ? gB : gA, (i & 1) ? gB : gA, (i & 1) ? gA : gB); }
static void l_gemm4(hipStream_t s, int i) { hipLaunchKernelGGL(gemm4_k, dim3(64), dim3(256), 0, s, (i & 1) ? gB : gA, (i & 1) ? gB : gA, (i & 1) ? gA : gB); }

int main()
{
    hipStream_t s;
    hipStreamCreate(&s);
    hipMalloc(&gA, SN * SN * 8);
    hipMalloc(&gB, SN * SN * 8);
    std::vector<double> h(SN * SN);
    for (int i = 0; i < SN * SN; ++i) h[i] = (i % 129 == 0) ? 1.0 : 1e-3 * ((i * 37) % 11);
    hipMemcpy(gA, h.data(), SN * SN * 8, hipMemcpyHostToDevice);
    hipMemcpy(gB, h.data(), SN * SN * 8, hipMemcpyHostToDevice);
    const int n = 100;
    printf("A empty<<<1,64>>>    eager %6.2f us/launch   graph %6.2f\n", time_chain(s, n, l_empty1), time_graph(s, n, l_empty1));
    printf("B empty<<<64,256>>>  eager %6.2f us/launch   graph %6.2f\n", time_chain(s, n, l_empty64), time_graph(s, n, l_empty64));
    printf("C gemm1 16x256       eager %6.2f us/launch   graph %6.2f\n", time_chain(s, n, l_gemm1), time_graph(s, n, l_gemm1));
    printf("D gemm4 64x256       eager %6.2f us/launch   graph %6.2f\n", time_chain(s, n, l_gemm4), time_graph(s, n, l_gemm4));

    printf("D2 gemm4 one XCD 512x256 eager %6.2f us/launch   graph %6.2f\n", time_chain(s, n, l_gemm4x), time_graph(s, n, l_gemm4x));
    unsigned *ctr;
    int *xcc;
    hipMalloc(&ctr, 4);
    hipMalloc(&xcc, 256 * 4);
    for (int mode = 0; mode < 3; ++mode) {
        const int stride = (mode == 1) ? 1 : 8, nw = (mode == 2) ? 16 : 32, grid = stride * nw;
        for (int phases : {1, 51, 101}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipMemcpy(gA, h.data(), SN * SN * 8, hipMemcpyHostToDevice);
                hipMemsetAsync(ctr, 0, 4, s);
                hipEventRecord(e0, s);
                hipLaunchKernelGGL(persist_k, dim3(grid), dim3(256), 0, s, gA, gB, ctr, phases, stride, nw, xcc);
                hipEventRecord(e1, s);
                hipStreamSynchronize(s);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            int hx[64];
            hipMemcpy(hx, xcc, nw * 4, hipMemcpyDeviceToHost);
            int hist[16] = {0};
            for (int i = 0; i < nw; ++i) hist[hx[i] & 15]++;
            printf("E persistent stride=%d nw=%d phases=%3d  total %8.2f us   xcc hist:", stride, nw, phases, best * 1e3f);
            for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
            printf("\n");
        }
    }
    hipError_t e = hipDeviceSynchronize();
    printf("final: %s\n", hipGetErrorString(e));
    return 0;
}
