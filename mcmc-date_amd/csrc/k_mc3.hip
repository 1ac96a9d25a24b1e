// k_mc3.hip -- the swap phase of Metropolis-coupled MCMC on the device (gfx950).
//
// Reference: `mc3 (MC3Settings (NChains 4) (SwapPeriod 2) (NSwaps 3))`, app/Main.hs:476-478; the algorithm itself lives in the
// package `mcmc` [external, not vendored]: every SwapPeriod iterations NSwaps adjacent temperature pairs of a group of NChains
// chains propose to exchange their states with probability min(1, exp((beta_i - beta_j)(ln pi(x_j) - ln pi(x_i)))),
// pi = prior x likelihood.  Here the TEMPERATURES move and the states stay where they are (the same Markov chain; 8 bytes per
// chain instead of a state), so the only thing a swap phase needs from other GPUs is the per-chain ln posterior -- the
// all-gather of SURVEY.md 8(e).
//
// Groups are NChains consecutive GLOBAL chains.  Every rank holds the temperature rank of every global chain and runs the
// whole swap phase on the gathered ln posteriors: the decisions are functions of global data and of counter-based random
// numbers keyed by (seed, global group, phase), so every rank arrives at the same table and a run does not depend on how the
// chains are dealt to GPUs.  One thread per group; a group is a handful of numbers.
#include "mh_device.hpp"

namespace mcd {

constexpr int kMc3MaxChains = 16;

// lnpost: [world][3][chains_per_rank] as mcd_shard_allgather leaves the ranks' [3][batch] posterior arrays (ln prior, ln
// likelihood, ln root-branch Jacobian); world = 1: the sampler's own array.
__global__ __launch_bounds__(256) void k_mc3_swap(Mc3Dev C, const double* __restrict__ lnpost, int world, int64_t per_rank, int n_swaps,
                                                 uint64_t seed, uint64_t phase, double* __restrict__ beta_local, int64_t chain0, int64_t batch)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = C.n_chains;
    if (g >= C.total / n) return;
    const int64_t base = g * n;
    double lnpi[kMc3MaxChains];
    int at[kMc3MaxChains];                                   // at[rank] = member (0 .. n - 1) that holds the rank
    int rk[kMc3MaxChains];
    for (int k = 0; k < n; ++k) {
        const int64_t c = base + k;
        const int64_t r = c / per_rank, i = c - r * per_rank;
        const double* p = lnpost + r * 3 * per_rank;
        lnpi[k] = p[i] + p[per_rank + i];
        rk[k] = C.rank[c];
        at[rk[k]] = k;
    }
    // n_swaps distinct adjacent pairs in random order: the head of a Fisher-Yates shuffle of the n - 1 pairs
    int pairs[kMc3MaxChains];
    for (int k = 0; k < n - 1; ++k) pairs[k] = k;
    const Rng rg = mh_rng(seed, g, phase);
    for (int j = 0; j < n_swaps; ++j) {
        double ua, ub;
        philox_block(rg, (uint32_t)j, ua, ub);
        int idx = j + (int)(ua * (double)(n - 1 - j));
        if (idx > n - 2) idx = n - 2;
        const int i = pairs[idx];
        pairs[idx] = pairs[j];
        pairs[j] = i;
        const int a = at[i], c = at[i + 1];
        const double log_r = (C.ladder[i] - C.ladder[i + 1]) * (lnpi[c] - lnpi[a]);
        atomicAdd(&C.tried[i], 1ull);
        if (log(ub) < log_r) {                               // NaN compares false: no swap
            rk[a] = i + 1;
            rk[c] = i;
            at[i] = c;
            at[i + 1] = a;
            atomicAdd(&C.accepted[i], 1ull);
        }
    }
    for (int k = 0; k < n; ++k) {
        const int64_t c = base + k;
        C.rank[c] = rk[k];
        if (c >= chain0 && c < chain0 + batch) beta_local[c - chain0] = C.ladder[rk[k]];
    }
}

hipError_t launch_mc3_swap(const Mc3Dev& C, const double* lnpost, int world, int64_t per_rank, int n_swaps, uint64_t seed, uint64_t phase,
                           double* beta_local, int64_t chain0, int64_t batch, hipStream_t st)
{
    const int64_t groups = C.total / C.n_chains;
    if (groups <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_mc3_swap, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, C, lnpost, world, per_rank, n_swaps, seed, phase,
                       beta_local, chain0, batch);
    return hipGetLastError();
}

}  // namespace mcd
