// wave_reduce.hpp -- fp64 cross-lane sums on the VALU only (DPP row rotations + v_permlane16/32_swap);
// __shfl_xor on a double lowers to two ds_bpermute_b32 (LDS-pipe latency) per step on gfx950.
#pragma once
#include <hip/hip_runtime.h>

namespace wr {

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// every lane of each row of 16 lanes receives the row sum
__device__ __forceinline__ double row16_allsum(double v)
{
    v += dpp_f64<0x128>(v);  // row_ror:8
    v += dpp_f64<0x124>(v);  // row_ror:4
    v += dpp_f64<0x122>(v);  // row_ror:2
    v += dpp_f64<0x121>(v);  // row_ror:1
    return v;
}
// every lane of each group of 32 lanes receives the group sum
__device__ __forceinline__ double group32_allsum(double v)
{
    v = row16_allsum(v);
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);  // odd rows of [0] <-> even rows of [1]
    const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
}
// every lane of the wave receives the wave sum
__device__ __forceinline__ double wave64_allsum(double v)
{
    v = group32_allsum(v);
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);  // upper half of [0] <-> lower half of [1]
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
}
// every lane of the wave receives the wave product (same data path as wave64_allsum)
__device__ __forceinline__ double wave64_allprod(double v)
{
    v *= dpp_f64<0x128>(v);
    v *= dpp_f64<0x124>(v);
    v *= dpp_f64<0x122>(v);
    v *= dpp_f64<0x121>(v);
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)rh[0], (int)rl[0]) * __hiloint2double((int)rh[1], (int)rl[1]);
    }
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) * __hiloint2double((int)rh[1], (int)rl[1]);
}
// every lane of the wave receives the wave maximum (same data path as wave64_allsum)
__device__ __forceinline__ double wave64_allmax(double v)
{
    v = fmax(v, dpp_f64<0x128>(v));
    v = fmax(v, dpp_f64<0x124>(v));
    v = fmax(v, dpp_f64<0x122>(v));
    v = fmax(v, dpp_f64<0x121>(v));
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = fmax(__hiloint2double((int)rh[0], (int)rl[0]), __hiloint2double((int)rh[1], (int)rl[1]));
    }
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return fmax(__hiloint2double((int)rh[0], (int)rl[0]), __hiloint2double((int)rh[1], (int)rl[1]));
}
__device__ __forceinline__ double bcast_lane(double v, int lane)  // wave-uniform `lane`
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

}  // namespace wr
