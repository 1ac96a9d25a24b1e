// k_nuts.hip -- the No-U-Turn sampler's tree building on the device, for B chains in lock step (gfx950).
// the reference's Hamiltonian proposal is `nuts` of the package `mcmc` (`nutsWith`,
// app/Hamiltonian.hs:95-105; dschrempf/mcmc 542c43f6, not vendored), restated from the algorithm it implements: Hoffman &
// Gelman, "The No-U-Turn Sampler", JMLR 15 (2014), Algorithm 3 -- slice variable, doubling in a random direction, U-turn
// and divergence stops, a uniform draw from the admissible leaves of the new sub tree, the biased progressive choice between
// the sub tree's candidate and the current proposal.
//
// The recursion of Algorithm 3 is unrolled into a per-chain state machine (doubling j, direction v, leaf index i inside the
// sub tree of 2^j leaves): every leapfrog step of every chain is one round of
//     [ k_nuts_step: finish the step, book the new leaf, load the edge to extend next, half kick + drift ]
//     [ k_prior_grad ] [ k_tree_grad ]                          (gradient of ln prior / ln likelihood at the new position)
// so a chain's tree needs no host decision.  The U-turn test of every aligned sub tree of 2^k leaves (what the recursion checks
// when two halves are joined) uses a stack of the sub trees' first leaves, one level per k; the candidate inside the new
// sub tree is chosen by reservoir sampling over its admissible leaves (the same uniform law as the recursion's pairwise
// choices, with other random numbers).  Chains that have finished wait for the others; the host only polls a counter.
//
// Random numbers: Philox4x32-10 as in the Metropolis-Hastings driver (mh_device.hpp), counter = (draw, chain, transition):
// draws 0x4000 + k / 2 -> momenta (Box-Muller pair), 1 -> slice, 0x10 + 2 j -> direction of doubling j, 0x11 + 2 j -> its
// merge, 0x100000 + leaf number -> reservoir.  tests/test_gpu_nuts.py holds the CPU twin that follows the same streams.
//
// One workgroup (256 threads) per chain; coordinates strided over the threads; dot products by workgroup reductions.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mh_device.hpp"
#include "mvn_kernels.h"

namespace mcd {

constexpr double NUTS_DELTA_MAX = 1000.0;

__device__ __forceinline__ double nuts_block_sum(double v, double* red)
{
    v = mh_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

// the entries of k_hmc.hip's hmc_grad_entry / hmc_state_slot, needed here as well
__device__ __forceinline__ double nuts_grad_entry(const HmcDev& D, int64_t b, int field, int v)
{
    const int64_t B = D.batch;
    const double* H = D.H + b * D.ld;
    const double* R = D.R + b * D.ld;
    const int l = 1, r = D.root_right;
    switch (field) {
        case 0: return D.gp_sc[0 * B + b];
        case 1: return D.gp_sc[1 * B + b];
        case 2: return D.gp_sc[2 * B + b] + D.gl_tH[b] - 1.0 / D.sc[2 * B + b];
        case 4: return D.gp_sc[3 * B + b] + D.gl_rMu[b] - 1.0 / D.sc[3 * B + b];
        case 5: return D.gp_sc[4 * B + b];
        default: break;
    }
    const double S = (H[0] - H[l]) * R[l] + (H[0] - H[r]) * R[r];
    if (field == 3) {
        double j = 0.0;
        if (v == l || v == r) j = R[v] / S;
        if (v == 0) j = -(R[l] + R[r]) / S;
        return D.gp_H[b * D.ld + v] + D.gl_H[b * D.ld + v] + j;
    }
    double j = 0.0;
    if (v == l || v == r) j = -(H[0] - H[v]) / S;
    return D.gp_R[b * D.ld + v] + D.gl_R[b * D.ld + v] + j;
}

__device__ __forceinline__ double* nuts_state_slot(const HmcDev& D, int64_t b, int field, int v)
{
    const int64_t B = D.batch;
    switch (field) {
        case 0: return D.sc + 0 * B + b;
        case 1: return D.sc + 1 * B + b;
        case 2: return D.sc + 2 * B + b;
        case 4: return D.sc + 3 * B + b;
        case 5: return D.sc + 4 * B + b;
        case 3: return D.H + b * D.ld + v;
        default: return D.R + b * D.ld + v;
    }
}

// (q_minus, r_minus, q_plus, r_plus) do not turn back on each other: (q+ - q-) . M^-1 r >= 0 at both ends (Algorithm 3)
__device__ __forceinline__ bool nuts_no_u_turn(const HmcDev& D, const double* qm, const double* rm, const double* qp, const double* rp, double* red)
{
    double a = 0.0, c = 0.0;
    for (int k = threadIdx.x; k < D.dim; k += blockDim.x) {
        const double d = qp[k] - qm[k];
        a += d * (rm[k] * D.inv_mass[k]);
        c += d * (rp[k] * D.inv_mass[k]);
    }
    a = nuts_block_sum(a, red);
    c = nuts_block_sum(c, red);
    return a >= 0.0 && c >= 0.0;
}

// the edge that is extended next goes to D.q / D.p / D.grad; half kick, drift, new position into the state arrays
__device__ __forceinline__ void nuts_launch_leaf(const HmcDev& D, const NutsDev& N, int64_t b, int v)
{
    const int64_t o = b * D.dim;
    const double* eq = (v < 0 ? N.qm : N.qp) + o;
    const double* ep = (v < 0 ? N.pm : N.pp) + o;
    const double* eg = (v < 0 ? N.gm : N.gp) + o;
    const double e = D.eps[b] * (double)v;
    for (int k = threadIdx.x; k < D.dim; k += blockDim.x) {
        const double g = eg[k];
        const double p = ep[k] + 0.5 * e * g;                       // p += eps / 2 grad
        const double q = eq[k] + e * D.inv_mass[k] * p;             // q += eps M^-1 p
        D.p[o + k] = p;
        D.q[o + k] = q;
        D.grad[o + k] = g;
        *nuts_state_slot(D, b, D.pos_field[k], D.pos_index[k]) = q;
    }
    if (threadIdx.x == 0) const_cast<double*>(D.dir)[b] = (double)v;
}

__device__ __forceinline__ int nuts_direction(uint64_t seed, int64_t chain, uint64_t transition, int j)
{
    double ua, ub;
    philox_block(mh_rng(seed, chain, transition), 0x10u + 2u * (uint32_t)j, ua, ub);
    return ua < 0.5 ? -1 : 1;
}

// Start of a transition: momenta, slice variable, the tree = the current point, first doubling, first leaf on its way.
// Needs D.q, D.grad, D.value of the current state (mcd_hmc_set_state / the previous transition).
__global__ __launch_bounds__(256) void k_nuts_begin(HmcDev D, NutsDev N, uint64_t seed, int64_t chain0, uint64_t transition)
{
    __shared__ double red[4];
    const int64_t b = blockIdx.x;
    const int64_t o = b * D.dim;
    const Rng g = mh_rng(seed, chain0 + b, transition);
    double kin = 0.0;
    for (int k = threadIdx.x; k < D.dim; k += blockDim.x) {
        double ua, ub;
        philox_block(g, 0x4000u + (uint32_t)(k >> 1), ua, ub);
        const double rad = sqrt(-2.0 * log(ua)), ang = 6.28318530717958647692 * ub;
        const double z = (k & 1) ? rad * sin(ang) : rad * cos(ang);
        const double p = z / sqrt(D.inv_mass[k]);                   // p ~ N(0, M)
        kin += p * p * D.inv_mass[k];
        const double q = D.q[o + k], gr = D.grad[o + k];
        N.qm[o + k] = q;
        N.qp[o + k] = q;
        N.pm[o + k] = p;
        N.pp[o + k] = p;
        N.gm[o + k] = gr;
        N.gp[o + k] = gr;
        N.qn[o + k] = q;
        N.gn[o + k] = gr;
    }
    kin = 0.5 * nuts_block_sum(kin, red);
    const double joint0 = D.value[b] - kin;
    double us, ub2;
    philox_block(g, 1u, us, ub2);
    const int v = nuts_direction(seed, chain0 + b, transition, 0);
    if (threadIdx.x == 0) {
        N.joint0[b] = joint0;
        N.log_u[b] = joint0 + log(us);
        N.lpn[b] = D.value[b];
        N.alpha[b] = 0.0;
        N.n_alpha[b] = 0;
        N.n[b] = 1;
        N.n1[b] = 0;
        N.s1[b] = 1;
        N.j[b] = 0;
        N.v[b] = v;
        N.i[b] = 0;
        N.leaf[b] = 0;
        N.depth[b] = 0;
        N.done[b] = 0;
    }
    __syncthreads();
    nuts_launch_leaf(D, N, b, v);
}

// One round: the gradient kernels have run at the position the drift reached.  Finish the leapfrog step (second half kick),
// book the leaf, decide, and send the next leaf on its way -- or mark the chain done.
__global__ __launch_bounds__(256) void k_nuts_step(HmcDev D, NutsDev N, uint64_t seed, int64_t chain0, uint64_t transition, int max_depth,
                                                   int* __restrict__ active)
{
    __shared__ double red[4];
    __shared__ int flag;
    const int64_t b = blockIdx.x;
    if (N.done[b]) return;                                          // workgroup-uniform
    const int64_t o = b * D.dim;
    const int v = N.v[b], j = N.j[b], i = N.i[b];
    const double e = D.eps[b] * (double)v;
    double* eq = (v < 0 ? N.qm : N.qp) + o;
    double* ep = (v < 0 ? N.pm : N.pp) + o;
    double* eg = (v < 0 ? N.gm : N.gp) + o;
    // ---- the new leaf: gradient from the kernels' outputs, second half kick, ln target; it becomes the tree's edge
    double kin = 0.0;
    for (int k = threadIdx.x; k < D.dim; k += blockDim.x) {
        const double g = nuts_grad_entry(D, b, D.pos_field[k], D.pos_index[k]);
        const double p = D.p[o + k] + 0.5 * e * g;
        kin += p * p * D.inv_mass[k];
        eq[k] = D.q[o + k];
        ep[k] = p;
        eg[k] = g;
    }
    kin = 0.5 * nuts_block_sum(kin, red);
    double lp;
    {
        const double* H = D.H + b * D.ld;
        const double* R = D.R + b * D.ld;
        const int l = 1, r = D.root_right;
        const double root_branch = D.sc[2 * D.batch + b] * D.sc[3 * D.batch + b] * ((H[0] - H[l]) * R[l] + (H[0] - H[r]) * R[r]);
        lp = D.lp[b] + D.ll[b] + log(1.0 / root_branch);           // prior x likelihood x jacobianRootBranch (app/Hamiltonian.hs:85-92)
    }
    double joint = lp - kin;
    if (!isfinite(joint)) joint = -INFINITY;                        // left the support / diverged: not admissible
    const double log_u = N.log_u[b];
    const bool nl = log_u <= joint;
    const bool sl = log_u < NUTS_DELTA_MAX + joint;
    int n1 = N.n1[b] + (nl ? 1 : 0);
    bool s1 = N.s1[b] != 0 && sl;
    const int leaf = N.leaf[b];
    if (nl) {                                                       // reservoir over the admissible leaves of the sub tree
        double ua, ub;
        philox_block(mh_rng(seed, chain0 + b, transition), 0x100000u + (uint32_t)leaf, ua, ub);
        if (ua * (double)n1 < 1.0) {
            for (int k = threadIdx.x; k < D.dim; k += blockDim.x) {
                N.qc[o + k] = eq[k];
                N.gc[o + k] = eg[k];
            }
            if (threadIdx.x == 0) N.lpc[b] = lp;
        }
    }
    if (threadIdx.x == 0) {
        const double d = joint - N.joint0[b];
        N.alpha[b] += fmin(1.0, exp(fmin(0.0, d)));
        N.n_alpha[b] += 1;
    }
    // ---- aligned sub trees: leaf i opens those of 2^k leaves with i % 2^k == 0 and closes those with (i + 1) % 2^k == 0
    for (int k = 1; k <= j; ++k) {
        const int size = 1 << k;
        double* lq = N.sq + ((int64_t)b * N.max_depth + (k - 1)) * D.dim;
        double* lr = N.sp + ((int64_t)b * N.max_depth + (k - 1)) * D.dim;
        if ((i & (size - 1)) == 0) {
            for (int c = threadIdx.x; c < D.dim; c += blockDim.x) {
                lq[c] = eq[c];
                lr[c] = ep[c];
            }
        } else if (((i + 1) & (size - 1)) == 0 && s1) {
            __syncthreads();
            // in time order: going forwards the first leaf is the minus end, going backwards the plus end
            const bool ok = v > 0 ? nuts_no_u_turn(D, lq, lr, eq, ep, red) : nuts_no_u_turn(D, eq, ep, lq, lr, red);
            s1 = s1 && ok;
        }
    }
    __syncthreads();
    // ---- decide
    int done = 0, jn = j, vn = v, in_ = i + 1, n = N.n[b];
    if (!s1) {
        done = 1;                                                   // the sub tree is abandoned, the proposal stays
        jn = j + 1;
    } else if (in_ == (1 << j)) {                                   // sub tree complete: biased progressive choice, U-turn of the whole tree
        double ua, ub;
        philox_block(mh_rng(seed, chain0 + b, transition), 0x11u + 2u * (uint32_t)j, ua, ub);
        if (n1 > 0 && ua * (double)n < (double)n1) {
            for (int k = threadIdx.x; k < D.dim; k += blockDim.x) {
                N.qn[o + k] = N.qc[o + k];
                N.gn[o + k] = N.gc[o + k];
            }
            if (threadIdx.x == 0) N.lpn[b] = N.lpc[b];
        }
        n += n1;
        __syncthreads();
        const bool s = nuts_no_u_turn(D, N.qm + o, N.pm + o, N.qp + o, N.pp + o, red);
        jn = j + 1;
        if (!s || jn >= max_depth) {
            done = 1;
        } else {
            vn = nuts_direction(seed, chain0 + b, transition, jn);
            in_ = 0;
            n1 = 0;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        N.n[b] = n;
        N.n1[b] = n1;
        N.s1[b] = s1 ? 1 : 0;
        N.j[b] = jn;
        N.v[b] = vn;
        N.i[b] = in_;
        N.leaf[b] = leaf + 1;
        N.done[b] = done;
        N.depth[b] = jn;
        if (!done) atomicAdd(active, 1);
        flag = done;
    }
    __syncthreads();
    if (!flag) nuts_launch_leaf(D, N, b, vn);
}

// End of a transition: the proposal becomes the chain's state (position, gradient and ln target with it)
__global__ __launch_bounds__(256) void k_nuts_end(HmcDev D, NutsDev N)
{
    const int64_t b = blockIdx.x;
    const int64_t o = b * D.dim;
    for (int k = threadIdx.x; k < D.dim; k += blockDim.x) {
        const double q = N.qn[o + k];
        D.q[o + k] = q;
        D.grad[o + k] = N.gn[o + k];
        *nuts_state_slot(D, b, D.pos_field[k], D.pos_index[k]) = q;
    }
    if (threadIdx.x == 0) D.value[b] = N.lpn[b];
}

hipError_t launch_nuts_begin(const HmcDev& D, const NutsDev& N, uint64_t seed, int64_t chain0, uint64_t transition, hipStream_t st)
{
    hipLaunchKernelGGL(k_nuts_begin, dim3((unsigned)D.batch), dim3(256), 0, st, D, N, seed, chain0, transition);
    return hipGetLastError();
}
hipError_t launch_nuts_step(const HmcDev& D, const NutsDev& N, uint64_t seed, int64_t chain0, uint64_t transition, int max_depth, int* active,
                            hipStream_t st)
{
    hipLaunchKernelGGL(k_nuts_step, dim3((unsigned)D.batch), dim3(256), 0, st, D, N, seed, chain0, transition, max_depth, active);
    return hipGetLastError();
}
hipError_t launch_nuts_end(const HmcDev& D, const NutsDev& N, hipStream_t st)
{
    hipLaunchKernelGGL(k_nuts_end, dim3((unsigned)D.batch), dim3(256), 0, st, D, N);
    return hipGetLastError();
}

}  // namespace mcd
