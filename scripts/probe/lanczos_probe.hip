// lanczos_probe.hip -- what would lambda_min of a 128 x 128 CoVO Hessian cost as a Lanczos iteration on ONE workgroup (no cross-workgroup
// hand-off at all), instead of the Chebyshev filter by squarings over 32 workgroups (sigma_ns.hip)?  Timing + accuracy probe, round 6.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I covo_mpc_amd/csrc scripts/probe/lanczos_probe.hip -o scripts/probe/lanczos_probe
//   python -c "import numpy as np; z=np.load('tests/golden/hessians_r03.npz'); np.concatenate([z[k] for k in z.files]).tofile('/tmp/h.bin')"
//   scripts/probe/lanczos_probe /tmp/h.bin 14
// Layout (512 threads): wave w owns the columns 16 w .. 16 w + 15; lane l holds rows l and l + 64 of that block (32 doubles).  The
// vector of the iteration is held by EVERY wave (rows l, l + 64 in two registers): a wave takes its 16 x's from its own registers by
// v_readlane, the eight column-block partials meet through LDS (one barrier per iteration, double buffered), and every wave forms
// the full sums, alpha, beta and the next vector redundantly -- no second barrier.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "wave_reduce.hpp"

constexpr int SN = 128, MAXM = 96;

__device__ __forceinline__ double rdlane(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rsq_(double x)  // hardware seed + 2 Newton steps
{
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    y = y * fma(-0.5 * x * y, y, 1.5);
    return y;
}

// out: [0] = iterations run, [1 .. MAXM] alpha, [1 + MAXM ..] beta, [1 + 2 MAXM ..] stamps (wall clock, 10 ns)
__global__ __launch_bounds__(512) void lanczos_kernel(const double *__restrict__ Aall, double *__restrict__ out_all, int iters)
{
    __shared__ double2 part[2][8][64];
    const double *A = Aall + (size_t)blockIdx.x * SN * SN;
    double *out = out_all + (size_t)blockIdx.x * (4 + 3 * MAXM + SN);
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long t0 = wall_clock64();
    double a0[16], a1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {  // A symmetric: A[l][16 w + j] = A[16 w + j][l] (coalesced in l)
        a0[j] = A[(size_t)(16 * w + j) * SN + l];
        a1[j] = A[(size_t)(16 * w + j) * SN + l + 64];
    }
    // start vector: fixed +-1 pattern (a hash of the index), normalised
    auto sgn = [](int i) { unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; return (h & 1u) ? 1.0 : -1.0; };
    double v0 = sgn(l) * 0.08838834764831845, v1 = sgn(l + 64) * 0.08838834764831845;  // 1/sqrt(128)
    double p0v = 0.0, p1v = 0.0, beta_prev = 0.0;
    const long long t1 = wall_clock64();
    int buf = 0;
    for (int m = 0; m < iters; ++m) {
        const double xs = (w >> 2) ? v1 : v0;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            const double xa = rdlane(xs, 16 * (w & 3) + j), xb = rdlane(xs, 16 * (w & 3) + j + 1);
            s0 = fma(a0[j], xa, s0);
            s1 = fma(a1[j], xa, s1);
            s2 = fma(a0[j + 1], xb, s2);
            s3 = fma(a1[j + 1], xb, s3);
        }
        part[buf][w][l] = make_double2(s0 + s2, s1 + s3);
        __syncthreads();
        double y0 = 0.0, y1 = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double2 p = part[buf][q][l];
            y0 += p.x;
            y1 += p.y;
        }
        buf ^= 1;
        const double alpha = wr::wave64_allsum(fma(v0, y0, v1 * y1));
        const double r0 = fma(-beta_prev, p0v, fma(-alpha, v0, y0)), r1 = fma(-beta_prev, p1v, fma(-alpha, v1, y1));
        const double b2 = wr::wave64_allsum(fma(r0, r0, r1 * r1));
        const double ib = rsq_(b2), beta = b2 * ib;
        p0v = v0; p1v = v1;
        v0 = r0 * ib; v1 = r1 * ib;
        beta_prev = beta;
        if (tid == 0) {
            out[1 + m] = alpha;
            out[1 + MAXM + m] = beta;
        }
    }
    const long long t2 = wall_clock64();
    if (tid == 0) {
        out[0] = (double)iters;
        out[1 + 2 * MAXM] = (double)(t1 - t0);
        out[2 + 2 * MAXM] = (double)(t2 - t1);
    }
}

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "/tmp/h.bin";
    const int nmat = argc > 2 ? atoi(argv[2]) : 14;
    std::vector<double> H((size_t)nmat * SN * SN);
    FILE *f = fopen(path, "rb");
    if (!f || fread(H.data(), sizeof(double), H.size(), f) != H.size()) { printf("cannot read %s\n", path); return 1; }
    fclose(f);
    for (int b = 0; b < nmat; ++b)
        for (int i = 0; i < SN; ++i)
            for (int j = 0; j < i; ++j) {
                const double s = 0.5 * (H[(size_t)b * SN * SN + i * SN + j] + H[(size_t)b * SN * SN + j * SN + i]);
                H[(size_t)b * SN * SN + i * SN + j] = H[(size_t)b * SN * SN + j * SN + i] = s;
            }
    double *dA, *dout;
    const size_t osz = 4 + 3 * MAXM + SN;
    hipMalloc(&dA, H.size() * 8);
    hipMalloc(&dout, (size_t)nmat * osz * 8);
    hipMemcpy(dA, H.data(), H.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int iters : {16, 32, 48, 64, 96}) {
        for (int grid : {1, 32}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(lanczos_kernel, dim3(grid == 1 ? 1 : std::min(grid, nmat)), dim3(512), 0, 0, dA, dout, iters);
                hipEventRecord(e1);
                hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = std::min(best, ms * 1e3f);
            }
            std::vector<double> o(osz);
            hipMemcpy(o.data(), dout, osz * 8, hipMemcpyDeviceToHost);
            printf("iters %2d grid %2d: launch %.2f us (events) | in-kernel: load A %.2f us, %d iterations %.2f us = %.3f us each\n", iters,
                   grid == 1 ? 1 : std::min(grid, nmat), best, o[1 + 2 * MAXM] / 100.0, iters, o[2 + 2 * MAXM] / 100.0,
                   o[2 + 2 * MAXM] / 100.0 / iters);
        }
    }
    // accuracy: smallest eigenvalue of T_m (host bisection) against the last (96-step) run's coefficients, all matrices
    hipLaunchKernelGGL(lanczos_kernel, dim3(nmat), dim3(512), 0, 0, dA, dout, MAXM);
    hipDeviceSynchronize();
    std::vector<double> o((size_t)nmat * osz);
    hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost);
    for (int b = 0; b < nmat; ++b) {
        const double *al = &o[(size_t)b * osz + 1], *be = &o[(size_t)b * osz + 1 + MAXM];
        auto theta1 = [&](int m) {
            double lo = -1e4, hi = 1e4;
            for (int it = 0; it < 200; ++it) {
                const double x = 0.5 * (lo + hi);
                int cnt = 0;
                double q = al[0] - x;
                if (q < 0) ++cnt;
                for (int i = 1; i < m; ++i) {
                    if (q == 0.0) q = 1e-300;
                    q = (al[i] - x) - be[i - 1] * be[i - 1] / q;
                    if (q < 0) ++cnt;
                }
                if (cnt >= 1) hi = x; else lo = x;
            }
            return 0.5 * (lo + hi);
        };
        const double ref = theta1(MAXM);
        printf("matrix %2d: theta_1(T_m) - theta_1(T_96):", b);
        for (int m : {12, 16, 20, 24, 32, 40, 48, 64}) printf("  m=%d %.1e", m, theta1(m) - ref);
        printf("   (theta_1 = %.12f)\n", ref);
    }
    return 0;
}
