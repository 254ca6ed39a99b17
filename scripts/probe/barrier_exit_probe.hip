// barrier_exit_probe.hip -- does s_barrier count only the SURVIVING waves of a workgroup?  (ISA: "if some waves in the threadgroup have
// already terminated, this waits on only the surviving waves".)  512-thread workgroups whose waves 4..7 return at entry; waves 0..3
// then pass 64 barriers exchanging data through LDS.   hipcc --offload-arch=gfx950 -O3 barrier_exit_probe.hip -o barrier_exit_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(int *out)
{
    __shared__ int buf[256];
    if (threadIdx.x >= 256) return;
    int v = threadIdx.x;
    for (int i = 0; i < 64; ++i) {
        buf[threadIdx.x] = v;
        __syncthreads();
        v = buf[(threadIdx.x + 1) & 255] + 1;
        __syncthreads();
    }
    out[blockIdx.x * 256 + threadIdx.x] = v;
}
int main()
{
    int *d, h[1024 * 256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1024), dim3(512), 0, 0, d);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 1024; ++b)
        for (int t = 0; t < 256; ++t) bad += h[b * 256 + t] != ((t + 64) & 255) + 64;
    printf("sync: %s, mismatches: %d\n", hipGetErrorString(e), bad);
    return bad != 0;
}
