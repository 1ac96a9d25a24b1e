// mh_inc_device.hpp -- LDS of the incremental ln likelihood of one chain by a workgroup of 256 threads (k_mh_inc.hip has the story):
// k_mh_step_wg (k_mh.hip) evaluates the sparse proposal it has just made.  dl: the current distances, then delta = x1 - x0; list: the
// moved rows, a quarter of the rows per wave (cnt4 entries each, at wave * NPz / 4); red: the waves' partial |z'|^2.
#pragma once
#include "mh_device.hpp"

namespace mcd {

constexpr int kIncThreads = 256;

struct IncShared {
    double dl[1024];
    int list[1024];
    int cnt4[4];
    double red[4];
};

__device__ __forceinline__ double inc_wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

}  // namespace mcd
