// rng.hip -- counter-based standard-normal fill (Philox4x32-10 + Box-Muller), gfx950.
//
// Stands in for jax.random.split + jax.random.normal inside jax.random.multivariate_normal
// (quadjax/controllers/covo.py:212-220, mppi.py:53-65).  jax's threefry bitstream is unpinned
// (setup.py:21) and unavailable here, so the stream is build-defined (rng_device.hpp); what is kept
// is the PROPERTY the sample-sharded step needs: element (global sample id, column) depends only on
// (key0, key1, id, column) -- never on the shard geometry.  The production step
// generates epsilon inside the noise kernels (noise_gemm.hip) with the same device function; this
// kernel materialises the identical values for parity tests and for callers that want epsilon.
#include "covo_common.hpp"
#include "rng_device.hpp"

__global__ __launch_bounds__(256) void randn_kernel(uint32_t k0, uint32_t k1, int64_t sample_offset, int n_samples,
                                                    int n_cols4, float4 *__restrict__ out)
{
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)n_samples * n_cols4;
    if (gid >= total) return;
    const uint32_t c4 = (uint32_t)(gid % n_cols4);
    const uint64_t id = (uint64_t)(sample_offset + (int64_t)(gid / n_cols4));
    out[gid] = rngd::normal4(c4, id, k0, k1);
}

int launch_randn(uint32_t k0, uint32_t k1, int64_t off, int n_samples, int n_cols, float *out, hipStream_t s)
{
    if (n_cols % 4 != 0) {
        covo_set_error("covo_randn: n_cols=%d must be a multiple of 4", n_cols);
        return COVO_E_BADARG;
    }
    const size_t total = (size_t)n_samples * (n_cols / 4);
    const int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(randn_kernel, dim3(grid), dim3(256), 0, s, k0, k1, off, n_samples, n_cols / 4,
                       reinterpret_cast<float4 *>(out));
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}
