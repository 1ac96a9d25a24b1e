// mh_prior_role.hpp -- the ln prior of a PROPOSED state of the two-launch Metropolis-Hastings step as a role of its own.
//
// k_mh_step (k_mh.hip) proposes; when asked not to evaluate the prior itself it leaves, per chain, the proposed state (H1, R1,
// sc1) and which blocks of the ln prior the proposal moved (pflags).  The ln prior and the ln likelihood of the proposal
// depend on nothing else, so they can be evaluated side by side: this role runs as extra workgroups of the tree-likelihood
// launch (k_tree_logpdf.hip, PRIOR variant).  One wave per chain, the same
// wave-level functions (prior_device.hpp) on the same numbers as the in-step evaluation: the same bits.
#pragma once
#include "mvn_kernels.h"
#include "prior_device.hpp"

namespace mcd {

// `hs`, `rs`: 2 n_nodes doubles of LDS private to the calling wave
__device__ __forceinline__ void mh_prior_role(const MhDev& M, const PriorDev& P, int64_t b, int lane, double* hs, double* rs)
{
    const int n = M.n_nodes;
    const int64_t B = M.batch;
    const int flags = M.pflags[b];
    double sc[5], pc[3];
#pragma unroll
    for (int i = 0; i < 5; ++i) sc[i] = M.sc1[i * B + b];
#pragma unroll
    for (int i = 0; i < 3; ++i) pc[i] = M.pcomp[b * 3 + i];
    if (flags != 0) {
        for (int w = lane; w < n; w += 64) {
            hs[w] = M.H1[b * M.ld + w];
            rs[w] = M.R1[b * M.ld + w];
        }
        __builtin_amdgcn_wave_barrier();
    }
    const double c0p = (flags & 1) ? prior_nodes_wave(P, lane, sc[2], hs) : pc[0];
    const double c1p = (flags & 2) ? prior_bd_wave(P, lane, sc[0], sc[1], hs) : pc[1];
    const double c2p = (flags & 4) ? prior_clock_wave(P, lane, sc[3], sc[4], hs, rs) : pc[2];
    if (lane == 0) {
        M.pcomp1[b * 3 + 0] = c0p;
        M.pcomp1[b * 3 + 1] = c1p;
        M.pcomp1[b * 3 + 2] = c2p;
        M.post1[b] = c0p + c1p + c2p;
    }
}

// The same with WPC = 2 or 4 waves per chain (a sampler's usual batch: every workgroup of both roles is resident at once, the launch
// lasts as long as its slowest wave): the chain's waves share the staging and deal the 64-node iterations of the birth-death
// and the clock block between them; the per-node summands go to LDS and the chain's first wave adds them lane by lane in the
// order of the iterations, then over the wave, exactly as prior_bd_wave / prior_clock_wave do alone -- the same bits.
// All waves of the workgroup call this (two workgroup barriers inside); `valid` = the chain exists.
// LDS of the chain: hs[n], rs[n], tb[NIT * 64], tc[NIT * 64], bc[8] with NIT = ceil((n - 1) / 64).
template <int WPC>
__device__ __forceinline__ void mh_prior_role_n(const MhDev& M, const PriorDev& P, int64_t b, bool valid, int sub, int lane, double* lds)
{
    const int n = M.n_nodes, NIT = (n - 1 + 63) >> 6;
    const int64_t B = M.batch;
    double* hs = lds;
    double* rs = hs + n;
    double* tb = rs + n;
    double* tc = tb + NIT * 64;
    double* bc = tc + NIT * 64;
    const int flags = valid ? M.pflags[b] : 0;
    double sc[5], pc[3];
#pragma unroll
    for (int i = 0; i < 5; ++i) sc[i] = M.sc1[i * B + b];
#pragma unroll
    for (int i = 0; i < 3; ++i) pc[i] = M.pcomp[b * 3 + i];
    if (flags != 0) {
        for (int w = sub * 64 + lane; w < n; w += 64 * WPC) {
            hs[w] = M.H1[b * M.ld + w];
            rs[w] = M.R1[b * M.ld + w];
        }
    }
    __syncthreads();                                       // the proposed state is in LDS
    if (flags & 2) {
        const bool near = prior_bd_near(sc[0], sc[1]);
        for (int it = sub; it < NIT; it += WPC) {
            const int v = 1 + lane + 64 * it;
            if (v < n) tb[it * 64 + lane] = prior_bd_term(P, v, near, sc[0], sc[1], hs);
        }
    }
    ClockCache cc{0.0, 0.0, 0.0, 0.0};
    if (flags & 4) {
        prior_clock_scalars(sc[4], cc);
        for (int it = sub; it < NIT; it += WPC) {
            const int v = 1 + lane + 64 * it;
            if (v < n) tc[it * 64 + lane] = prior_clock_term(P, v, sc[4], cc.lg_k, cc.log_t, hs, rs);
        }
    }
    if ((flags & 1) && sub == WPC - 1) {                   // the node priors: the chain's last wave (the first closes the sums)
        const double c0 = prior_nodes_wave(P, lane, sc[2], hs);
        if (lane == 0) bc[0] = c0;
    }
    __syncthreads();                                       // the summands are in LDS
    if (sub != 0 || !valid) return;
    const double c0p = (flags & 1) ? bc[0] : pc[0];
    double c1p = pc[1], c2p = pc[2];
    if (flags & 2) {
        double bd = 0.0;
        for (int it = 0; it < NIT; ++it)
            if (1 + lane + 64 * it < n) bd += tb[it * 64 + lane];
        c1p = prior_bd_finish(pr_wave_sum(bd), sc[0], sc[1]);
    }
    if (flags & 4) {
        double clock = 0.0;
        for (int it = 0; it < NIT; ++it)
            if (1 + lane + 64 * it < n) clock += tc[it * 64 + lane];
        c2p = prior_clock_finish(P, pr_wave_sum(clock), sc[3], sc[4], cc.hyper);
    }
    if (lane == 0) {
        M.pcomp1[b * 3 + 0] = c0p;
        M.pcomp1[b * 3 + 1] = c1p;
        M.pcomp1[b * 3 + 2] = c2p;
        M.post1[b] = c0p + c1p + c2p;
    }
}

__device__ __forceinline__ void mh_prior_role2(const MhDev& M, const PriorDev& P, int64_t b, bool valid, int sub, int lane, double* lds)
{
    mh_prior_role_n<2>(M, P, b, valid, sub, lane, lds);
}

// doubles of LDS per chain for mh_prior_role_n
__host__ __device__ inline size_t mh_prior_role2_doubles(int n_nodes) { return 2 * (size_t)n_nodes + 2 * (size_t)((n_nodes - 1 + 63) / 64) * 64 + 8; }

}  // namespace mcd
