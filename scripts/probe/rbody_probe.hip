// rbody_probe.hip -- cost of the reward stage's per-step body, as a rolled loop, with pieces removed one at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
constexpr int ITER = 512;
constexpr float LN2 = 0.69314718056f;

__device__ __forceinline__ float atan2abs6(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = __builtin_fmaxf(__builtin_fmaxf(ax, ay), 1e-30f), mn = __builtin_fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(mx);
    const float s = t * t;
    float r = 0.007374854292720556f;
    r = __builtin_fmaf(r, s, -0.03552231565117836f);
    r = __builtin_fmaf(r, s, 0.08217037469148636f);
    r = __builtin_fmaf(r, s, -0.13398927450180054f);
    r = __builtin_fmaf(r, s, 0.1986188441514969f);
    r = __builtin_fmaf(r, s, -0.33325397968292236f);
    r = r * s;
    r = __builtin_fmaf(r, t, t);
    r = (ay > ax) ? (1.57079637f - r) : r;
    r = (x < 0.0f) ? (3.14159274f - r) : r;
    return r;
}
__device__ __forceinline__ float sat01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }

// MASK bits: 1 sqrt x2, 2 atan, 4 log + clamps, 8 freeze/accumulate
template <int MASK>
__global__ __launch_bounds__(256) void probe(float *out, const float *in, unsigned long long *ticks)
{
    float e0 = in[threadIdx.x], e1 = in[threadIdx.x + 256], e2 = in[threadIdx.x + 512], e3 = in[threadIdx.x + 768];
    float acc = 0.f, r_before = 0.f;
    bool done_before = false;
    const unsigned long long r0 = wall_clock64();
#pragma unroll 2
    for (int it = 0; it < ITER; ++it) {
        asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
        float err_pos = e0, err_vel = e1;
        if (MASK & 1) { err_pos = __builtin_amdgcn_sqrtf(e0); err_vel = __builtin_amdgcn_sqrtf(__builtin_fabsf(e1)); }
        float yaw = e2;
        if (MASK & 2) yaw = atan2abs6(e2, e3);
        float r = __builtin_fmaf(err_vel, -0.05f, 1.3f);
        r = __builtin_fmaf(err_pos, -0.4f, r);
        if (MASK & 4) {
            const float l2 = __builtin_amdgcn_logf(err_pos + 1.0f);
            r = __builtin_fmaf(sat01(l2 * (4.0f * LN2)), -0.4f, r);
            r = __builtin_fmaf(sat01(l2 * (8.0f * LN2)), -0.2f, r);
            r = __builtin_fmaf(sat01(l2 * (16.0f * LN2)) + sat01(l2 * (32.0f * LN2)), -0.1f, r);
        }
        r = __builtin_fmaf(yaw, -0.2f, r);
        if (MASK & 8) {
            const bool done = __float_as_int(e1) < 0;
            r = done_before ? r_before : r;
            done_before = done_before | done;
            r_before = r;
        }
        acc += r;
        e0 += 1e-3f; e1 += 1e-3f; e2 -= 1e-3f; e3 += 2e-3f;
    }
    const unsigned long long r1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = r1 - r0;
}

template <int MASK>
void run(const char *name)
{
    float *out, *in;
    unsigned long long *ticks;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMalloc(&in, 4096);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0.01f + 0.001f * i;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    hipMalloc(&ticks, 256 * 8 * 4 * 8);
    for (int W : {1, 2, 3, 4, 6}) {
        const int grid = 256 * W;
        hipLaunchKernelGGL(probe<MASK>, dim3(grid), dim3(256), 0, 0, out, in, ticks);
        hipLaunchKernelGGL(probe<MASK>, dim3(grid), dim3(256), 0, 0, out, in, ticks);
        hipDeviceSynchronize();
        std::vector<unsigned long long> t(grid * 4);
        hipMemcpy(t.data(), ticks, t.size() * 8, hipMemcpyDeviceToHost);
        std::sort(t.begin(), t.end());
        const double ns = (double)t[t.size() / 2] * 10.0 / ITER;
        printf("%-40s W=%d  ns/step/wave %7.2f   ns/step/SIMD %7.2f\n", name, W, ns, ns / W);
    }
    hipFree(out); hipFree(in); hipFree(ticks);
}

int main()
{
    run<15>("full R body");
    run<14>("without the two sqrt");
    run<13>("without atan");
    run<11>("without log + clamps");
    run<7>("without freeze");
    run<0>("only the fma skeleton");
    run<2>("atan only");
    run<4>("log + clamps only");
    return 0;
}
