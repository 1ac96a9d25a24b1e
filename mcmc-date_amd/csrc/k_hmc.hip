// k_hmc.hip -- leapfrog integrator of the Hamiltonian proposal on the device (gfx950).  SURVEY.md 8(f) row f3, second part.
//
// Potential: U(q) = -ln [ prior x likelihood x jacobianRootBranch ](state(q))       (`htargetWith`, app/Hamiltonian.hs:72-92)
// Position:  q = the masked, reversed fold of the state record (`toVector`, `getMask`, app/Hamiltonian.hs:33-60):
//            pos_field[i] in {0 birth, 1 death, 2 tH, 3 height, 4 rMu, 5 rVar, 6 rate}, pos_index[i] = node id.
// One leapfrog step (step size eps_b per chain, diagonal inverse masses):
//     p += eps/2 grad(q);   q += eps Minv p;   p += eps/2 grad(q)
// The gradient comes from the batched kernels (k_prior_grad.hip for the prior, k_tree_grad.hip for the likelihood) plus
// the five-number Jacobian term evaluated here; this file holds the element-wise part: assemble the gradient in the
// position layout, kick, drift and scatter the new position into the state arrays.  One thread per (chain, coordinate).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mvn_kernels.h"

namespace mcd {

// d ln target / d q_i for chain b, from the outputs of the two gradient kernels + the Jacobian factor 1 / rootBranch,
// rootBranch = tH rMu (t_l r_l + t_r r_r)                                             (app/Probability.hs:393-410)
__device__ __forceinline__ double hmc_grad_entry(const HmcDev& D, int64_t b, int field, int v)
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
        if (v == l || v == r) j = R[v] / S;                       // d/d h_v of -ln S
        if (v == 0) j = -(R[l] + R[r]) / S;
        return D.gp_H[b * D.ld + v] + D.gl_H[b * D.ld + v] + j;
    }
    double j = 0.0;
    if (v == l || v == r) j = -(H[0] - H[v]) / S;
    return D.gp_R[b * D.ld + v] + D.gl_R[b * D.ld + v] + j;
}

__device__ __forceinline__ double* hmc_state_slot(const HmcDev& D, int64_t b, int field, int v)
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

// p += kick * eps_b * dir_b * grad.  `dir` (+1 / -1 per chain, may be null = +1) integrates backwards in time (NUTS-style
// doubling).  The drift is a separate launch (k_hmc_drift): the Jacobian term of one coordinate reads state entries that
// belong to other coordinates, so every gradient entry is read before any position is written.
// use_pos_grad != 0: the gradient is taken from D.grad (position layout, supplied by the caller with the position)
// instead of being assembled from the outputs of the gradient kernels
__global__ __launch_bounds__(256) void k_hmc_kick(HmcDev D, double kick, int use_pos_grad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.batch * D.dim) return;
    const int64_t b = i / D.dim;
    const int k = (int)(i - b * D.dim);
    const int field = D.pos_field[k], v = D.pos_index[k];
    const double e = D.eps[b] * (D.dir ? D.dir[b] : 1.0);
    const double g = use_pos_grad ? D.grad[i] : hmc_grad_entry(D, b, field, v);
    D.p[i] = D.p[i] + kick * e * g;
    D.grad[i] = g;
}

// the state arrays receive the position D.q (masked-out entries keep their values)
__global__ __launch_bounds__(256) void k_hmc_scatter(HmcDev D)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.batch * D.dim) return;
    const int64_t b = i / D.dim;
    const int k = (int)(i - b * D.dim);
    *hmc_state_slot(D, b, D.pos_field[k], D.pos_index[k]) = D.q[i];
}

__global__ __launch_bounds__(256) void k_hmc_drift(HmcDev D)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.batch * D.dim) return;
    const int64_t b = i / D.dim;
    const int k = (int)(i - b * D.dim);
    const int field = D.pos_field[k], v = D.pos_index[k];
    const double e = D.eps[b] * (D.dir ? D.dir[b] : 1.0);
    double* slot = hmc_state_slot(D, b, field, v);
    const double q = *slot + e * D.inv_mass[k] * D.p[i];
    *slot = q;
    D.q[i] = q;
}

// value[b] = ln prior + ln likelihood + ln jacobianRootBranch of the current state; q and grad in the position layout
__global__ __launch_bounds__(256) void k_hmc_collect(HmcDev D)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.batch * D.dim) return;
    const int64_t b = i / D.dim;
    const int k = (int)(i - b * D.dim);
    const int field = D.pos_field[k], v = D.pos_index[k];
    D.q[i] = *hmc_state_slot(D, b, field, v);
    D.grad[i] = hmc_grad_entry(D, b, field, v);
    if (k == 0) {
        const double* H = D.H + b * D.ld;
        const double* R = D.R + b * D.ld;
        const int l = 1, r = D.root_right;
        const double root_branch = D.sc[2 * D.batch + b] * D.sc[3 * D.batch + b] * ((H[0] - H[l]) * R[l] + (H[0] - H[r]) * R[r]);
        D.value[b] = D.lp[b] + D.ll[b] + log(1.0 / root_branch);
    }
}

// kick and drift of one leapfrog step in ONE launch: a workgroup per chain, so that a barrier separates "every gradient entry of
// the chain has been read" (the Jacobian term of a coordinate reads state entries that belong to other coordinates) from "the
// new position is written into the state".  p += kick eps g;  q += eps M^-1 p.
__global__ __launch_bounds__(256) void k_hmc_kick_drift(HmcDev D, double kick)
{
    const int64_t b = blockIdx.x;
    const int64_t o = b * D.dim;
    const double e = D.eps[b] * (D.dir ? D.dir[b] : 1.0);
    for (int k = threadIdx.x; k < D.dim; k += blockDim.x) {
        const double g = hmc_grad_entry(D, b, D.pos_field[k], D.pos_index[k]);
        D.grad[o + k] = g;
        D.p[o + k] = D.p[o + k] + kick * e * g;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < D.dim; k += blockDim.x) {
        double* slot = hmc_state_slot(D, b, D.pos_field[k], D.pos_index[k]);
        const double q = *slot + e * D.inv_mass[k] * D.p[o + k];
        *slot = q;
        D.q[o + k] = q;
    }
}

// the closing half kick together with k_hmc_collect
__global__ __launch_bounds__(256) void k_hmc_kick_collect(HmcDev D, double kick)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.batch * D.dim) return;
    const int64_t b = i / D.dim;
    const int k = (int)(i - b * D.dim);
    const int field = D.pos_field[k], v = D.pos_index[k];
    const double e = D.eps[b] * (D.dir ? D.dir[b] : 1.0);
    const double g = hmc_grad_entry(D, b, field, v);
    D.p[i] = D.p[i] + kick * e * g;
    D.grad[i] = g;
    D.q[i] = *hmc_state_slot(D, b, field, v);
    if (k == 0) {
        const double* H = D.H + b * D.ld;
        const double* R = D.R + b * D.ld;
        const int l = 1, r = D.root_right;
        const double root_branch = D.sc[2 * D.batch + b] * D.sc[3 * D.batch + b] * ((H[0] - H[l]) * R[l] + (H[0] - H[r]) * R[r]);
        D.value[b] = D.lp[b] + D.ll[b] + log(1.0 / root_branch);
    }
}

static unsigned hmc_grid(const HmcDev& D) { return (unsigned)((D.batch * D.dim + 255) / 256); }

hipError_t launch_hmc_kick(const HmcDev& D, double kick, int use_pos_grad, hipStream_t st)
{
    hipLaunchKernelGGL(k_hmc_kick, dim3(hmc_grid(D)), dim3(256), 0, st, D, kick, use_pos_grad);
    return hipGetLastError();
}
hipError_t launch_hmc_scatter(const HmcDev& D, hipStream_t st)
{
    hipLaunchKernelGGL(k_hmc_scatter, dim3(hmc_grid(D)), dim3(256), 0, st, D);
    return hipGetLastError();
}
hipError_t launch_hmc_drift(const HmcDev& D, hipStream_t st)
{
    hipLaunchKernelGGL(k_hmc_drift, dim3(hmc_grid(D)), dim3(256), 0, st, D);
    return hipGetLastError();
}
hipError_t launch_hmc_kick_drift(const HmcDev& D, double kick, hipStream_t st)
{
    hipLaunchKernelGGL(k_hmc_kick_drift, dim3((unsigned)D.batch), dim3(256), 0, st, D, kick);
    return hipGetLastError();
}
hipError_t launch_hmc_kick_collect(const HmcDev& D, double kick, hipStream_t st)
{
    hipLaunchKernelGGL(k_hmc_kick_collect, dim3(hmc_grid(D)), dim3(256), 0, st, D, kick);
    return hipGetLastError();
}
hipError_t launch_hmc_collect(const HmcDev& D, hipStream_t st)
{
    hipLaunchKernelGGL(k_hmc_collect, dim3(hmc_grid(D)), dim3(256), 0, st, D);
    return hipGetLastError();
}

}  // namespace mcd
