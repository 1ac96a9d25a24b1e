// mh_inc_device.hpp -- the incremental ln likelihood of one chain by a workgroup of 256 threads (k_mh_inc.hip has the story): shared by
// k_mh_step_wg (k_mh.hip), which evaluates the sparse proposal it has just made, and nothing else.
#pragma once
#include "mh_device.hpp"

namespace mcd {

constexpr int kIncThreads = 256;

struct IncShared {
    double dl[1024];
    int list[1024];
    int count;
    double red[4];
};

__device__ __forceinline__ double inc_wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// z' = z + sum_j delta_j W[:, j] over the rows j whose distance moved (delta = x1 - x0, row order), z' -> zprop, returns
// c - 1/2 (logdet + |z'|^2) in thread 0 (all 256 threads of the workgroup call it; it synchronises the workgroup)
__device__ __forceinline__ double mh_inc_ll_block(const MvnDev& V, const MhInc& I, const double* __restrict__ x1, const double* __restrict__ x0,
                                                  const double* __restrict__ zc, double* __restrict__ zo, IncShared& sh, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int n = V.n, NP = I.NPz;
    for (int j = tid; j < NP; j += kIncThreads) sh.dl[j] = (j < n) ? x1[j] - x0[j] : 0.0;
    __syncthreads();
    if (wave == 0) {                                         // the moved rows in row order (NaN != 0: kept, and the NaN then reaches q)
        int c = 0;
        for (int j0 = 0; j0 < NP; j0 += 64) {
            const bool mv = sh.dl[j0 + lane] != 0.0;
            const uint64_t mk = __builtin_amdgcn_ballot_w64(mv);
            if (mv) sh.list[c + __builtin_popcountll(mk & ((1ull << lane) - 1ull))] = j0 + lane;
            c += __builtin_popcountll(mk);
        }
        if (lane == 0) sh.count = c;
    }
    __syncthreads();
    const int cnt = sh.count;
    double zp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) zp[k] = (tid + kIncThreads * k < NP) ? zc[tid + kIncThreads * k] : 0.0;
    for (int m0 = 0; m0 < cnt; m0 += 4) {                    // four columns in flight
        double w[4][4], d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = (m0 + u < cnt) ? m0 + u : cnt - 1;
            const int j = sh.list[m];
            d[u] = (m0 + u < cnt) ? sh.dl[j] : 0.0;
            const double* wc = V.Wc + (size_t)j * NP;
#pragma unroll
            for (int k = 0; k < 4; ++k) w[u][k] = (tid + kIncThreads * k < NP) ? wc[tid + kIncThreads * k] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) zp[k] = fma(d[u], w[u][k], zp[k]);
    }
    double sq = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (tid + kIncThreads * k < NP) zo[tid + kIncThreads * k] = zp[k];
        sq = fma(zp[k], zp[k], sq);
    }
    sq = inc_wave_sum(sq);
    if (lane == 0) sh.red[wave] = sq;
    __syncthreads();
    const double q = ((sh.red[0] + sh.red[1]) + sh.red[2]) + sh.red[3];
    return V.c + (-0.5) * (V.logdet + q);                    // :169
}

}  // namespace mcd
