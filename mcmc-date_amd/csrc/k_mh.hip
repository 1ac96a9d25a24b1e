// k_mh.hip -- lock-step Metropolis-Hastings-Green steps for a batch of independent chains (gfx950).
// SURVEY.md 8(f) row f2.  All chains execute the same proposal of the cycle at the same time, each with its own
// counter-based random numbers, tuning parameter and accept/reject decision; the posterior of the proposed states
// is evaluated by the batched prior and likelihood kernels between `mh_propose` and `mh_accept`.
//
// Proposals restated here (paths relative to the reference):
//   slide node / scale sub tree / pulley, ultrametric   lib/Mcmc/Tree/Proposal/Ultrametric.hs:50-59, 126-149, 221-286
//   truncatedNormalSample                               lib/Mcmc/Tree/Proposal/Internal.hs:107-138
//   truncated normal density / quantile                 lib/Statistics/Distribution/TruncatedNormal.hs:55-130
//   scale branch / scale (sub) tree of rates            lib/Mcmc/Tree/Proposal/Unconstrained.hs:40-130
//   scaleNormAndTreeContrarily                          .../Unconstrained.hs:221-256
//   scaleVarianceAndTree (and the autocorrelated form)  .../Unconstrained.hs:286-316, 354-386
//   scaleUnbiased / scaleContrarily, MHG acceptance, auto tuning: package `mcmc` [external, dschrempf/mcmc 542c43f6]
//   liftProposalWith jacobianRootBranch                 app/Definitions.hs:148 ff., app/Probability.hs:393-410
//
// Mapping: one wave per chain, lanes = nodes (strided for trees with more than 64 nodes).  Pre-order numbering makes
// every sub tree a contiguous index range [v, v + size[v]), so "scale the sub tree" is a range test per lane.  The
// scalar part of a proposal (random draw, bounds, ratio) is computed redundantly by all lanes of the wave.
// Random numbers: Philox4x32-10, counter = (draw, chain, step_lo, step_hi), key = seed; the proposal's draws use
// draw = 0, 1, ..., the acceptance uniform is draw 0xFFFFFFFF, the boost of a gamma with shape < 1 is 0xFFFFFFFE.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mh_device.hpp"
#include "prior_device.hpp"

namespace mcd {

// Proposes for every chain and evaluates the ln prior of the proposed state in the same launch: the proposed heights and
// rates are staged in LDS (one region per wave), the prior reads them there, then they go to global memory for the
// likelihood kernel.  Writes sc1, H1, R1, lnqj (ln q-ratio * Jacobian without the root-branch factor) and post1[0] = ln prior.
__global__ __launch_bounds__(256) void k_mh_propose(MhDev M, PriorDev P, const int32_t* __restrict__ sched, int64_t sched_idx,
                                                    uint64_t step, uint64_t seed)
{
    extern __shared__ double sh[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (b >= M.batch) return;                             // wave-uniform; no workgroup barriers in this kernel
    const int n = M.n_nodes;
    double* Hs = sh + (size_t)wave * 2 * n;
    double* Rs = Hs + n;
    const int p = sched[sched_idx];
    const int64_t B = M.batch;
    double sc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) sc[i] = M.sc[i * B + b];
    const double t = M.tune[b * M.n_prop + p];
    const StepDraws dr = mh_step_draws(M, p, t, mh_rng(seed, M.chain0 + b, step));
    const double lnqj = mh_propose_wave(M, p, t, dr, lane, sc, M.H + b * M.ld, M.R + b * M.ld, Hs, Rs);
    __builtin_amdgcn_wave_barrier();
    const double lp1 = prior_eval_wave(P, lane, sc[0], sc[1], sc[2], sc[3], sc[4], Hs, Rs, nullptr);
    for (int w = lane; w < n; w += 64) {
        M.H1[b * M.ld + w] = Hs[w];
        M.R1[b * M.ld + w] = Rs[w];
    }
    if (lane < 5) {
        double mine = sc[0];
#pragma unroll
        for (int i = 1; i < 5; ++i)
            if (lane == i) mine = sc[i];
        M.sc1[lane * B + b] = mine;
    }
    if (lane == 0) {
        M.lnqj[b] = lnqj;
        M.post1[b] = lp1;
    }
}

// Accept or reject; post = (ln prior, ln likelihood, ln jacobianRootBranch), [3][batch].
__global__ __launch_bounds__(256) void k_mh_accept(MhDev M, const int32_t* __restrict__ sched, int64_t sched_idx,
                                                   uint64_t step, uint64_t seed, double* __restrict__ trace_alpha,
                                                   int8_t* __restrict__ trace_accept)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= M.batch) return;
    const int p = sched[sched_idx];
    const int64_t B = M.batch;
    const double lp = M.post[b], ll = M.post[B + b], lj = M.post[2 * B + b];
    const double lp1 = M.post1[b], ll1 = M.post1[B + b], lj1 = M.post1[2 * B + b];
    double la = (lp1 + ll1) - (lp + ll) + M.lnqj[b];
    if (M.jac_root[p]) la += lj1 - lj;
    double ua, ub;
    philox_block(mh_rng(seed, M.chain0 + b, step), 0xFFFFFFFFu, ua, ub);
    const bool ok = (la >= 0) || (ua < exp(la));
    if (ok) {
        const double* H1 = M.H1 + b * M.ld;
        const double* R1 = M.R1 + b * M.ld;
        double* H = M.H + b * M.ld;
        double* R = M.R + b * M.ld;
        for (int w = lane; w < M.n_nodes; w += 64) {
            H[w] = H1[w];
            R[w] = R1[w];
        }
        if (lane < 5) M.sc[lane * B + b] = M.sc1[lane * B + b];
        if (lane < 3) M.post[lane * B + b] = M.post1[lane * B + b];
    }
    if (lane == 0) {
        const int64_t i = b * M.n_prop + p;
        M.tried[i] += 1;
        if (ok) M.acc[i] += 1;
        if (trace_alpha) trace_alpha[b] = la;
        if (trace_accept) trace_accept[b] = ok ? 1 : 0;
    }
}

// After every iteration: running sums of the absolute node ages tH * h_v (the quantity the reference's
// node-age summaries report, scripts/trees-monitor-summary-ultrametric).
__global__ __launch_bounds__(256) void k_mh_accumulate(MhDev M)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M.batch * M.n_nodes) return;
    const int64_t b = i / M.n_nodes;
    const int v = (int)(i - b * M.n_nodes);
    const double a = M.sc[2 * M.batch + b] * M.H[b * M.ld + v];
    M.age_sum[b * M.n_nodes + v] += a;
    M.age_sq[b * M.n_nodes + v] += a * a;
}

// mcmc's auto tuning [external]: t' = clamp(t exp(2 (rate - optimal(dim))), 1e-5, 1e3); counters reset.
__global__ __launch_bounds__(256) void k_mh_tune(MhDev M)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M.batch * M.n_prop) return;
    const int n_tried = M.tried[i];
    if (n_tried > 0) {
        const int dim = M.dim[i % M.n_prop];
        const double opt = mh_optimal_rate(dim);
        const double r = (double)M.acc[i] / (double)n_tried;
        double t = M.tune[i] * exp(2.0 * (r - opt));
        t = (t < 1e-5) ? 1e-5 : (t > 1e3) ? 1e3 : t;
        M.tune[i] = t;
    }
    M.acc[i] = 0;
    M.tried[i] = 0;
}

hipError_t launch_mh_propose(const MhDev& M, const PriorDev& P, const int32_t* sched, int64_t sched_idx, uint64_t step, uint64_t seed,
                             hipStream_t st)
{
    const size_t per_wave = sizeof(double) * 2 * (size_t)M.n_nodes;
    int wpb = 4;
    while (wpb > 1 && per_wave * wpb > 60 * 1024) wpb >>= 1;
    if (per_wave * wpb > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_mh_propose, dim3((unsigned)((M.batch + wpb - 1) / wpb)), dim3(64 * wpb), per_wave * wpb, st, M, P, sched, sched_idx,
                       step, seed);
    return hipGetLastError();
}
hipError_t launch_mh_accept(const MhDev& M, const int32_t* sched, int64_t sched_idx, uint64_t step, uint64_t seed,
                            double* trace_alpha, int8_t* trace_accept, hipStream_t st)
{
    const int wpb = 4;
    hipLaunchKernelGGL(k_mh_accept, dim3((unsigned)((M.batch + wpb - 1) / wpb)), dim3(64 * wpb), 0, st, M, sched, sched_idx, step, seed,
                       trace_alpha, trace_accept);
    return hipGetLastError();
}
hipError_t launch_mh_accumulate(const MhDev& M, hipStream_t st)
{
    const int64_t n = M.batch * M.n_nodes;
    hipLaunchKernelGGL(k_mh_accumulate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, M);
    return hipGetLastError();
}
hipError_t launch_mh_tune(const MhDev& M, hipStream_t st)
{
    const int64_t n = M.batch * M.n_prop;
    hipLaunchKernelGGL(k_mh_tune, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, M);
    return hipGetLastError();
}

}  // namespace mcd
