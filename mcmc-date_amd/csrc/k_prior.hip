// k_prior.hip -- batched log prior of McmcDate's state (gfx950).  SURVEY.md 8(f) row f1.
//
// priorFunction ht md cb cs bs x  (app/Probability.hs:127-150)
//   = calibrateConstrainBraceSoft tH cb cs bs timeTree                      (Prior/Node/Combined.hs:70-92)
//   * exponential 1 birth * exponential 1 death * birthDeath ConditionOnTimeOfMrca birth death 1 t'   (:66-85)
//   * exponential ht rMu * gamma (3/2) (1/6) rVar * <relaxed clock model> 1 rVar t' rateTree           (:96-124)
// with t' = heightTreeToLengthTree timeTree.  All factors are evaluated in the log domain.
//
// Mapping: one wave per chain, lanes = nodes (strided when the tree has more than 64 nodes).  Every
// per-node term is independent.  The only recursion of the reference, E at the bottom of a branch in
// birthDeathWith (Prior/BirthDeath.hs:186-239), follows the chain of FIRST children down to a tip; it
// composes the flow of Stadler's Eq. [1] branch by branch, and that flow is a semigroup, so for a node
// at relative height t (tips at 0, rho = 1) E = mu (1 - x) / (la - mu x), x = exp(-(la - mu) t), in one
// step.  (Whenever a branch is <= 0 the reference's guard makes the whole prior 0 anyway.)  In the
// near-critical regime |la - mu| < 1e-6 the reference switches to a first-order formula that is NOT a
// semigroup; there each lane walks the chain exactly as the reference does.
// One DPP/readlane reduction per component at the end.  O(n_nodes) work per chain: negligible next to
// the likelihood sweep; this kernel exists so that a full posterior can be evaluated without leaving
// the device.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mvn_kernels.h"
#include "prior_device.hpp"

namespace mcd {

__global__ void __launch_bounds__(256) k_prior(PriorDev P, const double* __restrict__ birth, const double* __restrict__ death,
                                               const double* __restrict__ tH, const double* __restrict__ H,
                                               const double* __restrict__ rMu, const double* __restrict__ rVar,
                                               const double* __restrict__ Rt, int64_t lds, int64_t batch,
                                               double* __restrict__ lp, double* __restrict__ comp)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= batch) return;                                   // wave-uniform; no barriers in this kernel
    double c[3];
    const double total = prior_eval_wave(P, lane, birth[b], death[b], tH[b], rMu[b], rVar[b], H + b * lds, Rt + b * lds, c);
    if (lane == 0) {
        lp[b] = total;
        if (comp) {
            comp[b * 3 + 0] = c[0];
            comp[b * 3 + 1] = c[1];
            comp[b * 3 + 2] = c[2];
        }
    }
}

hipError_t launch_prior(const PriorDev& P, const double* birth, const double* death, const double* tH, const double* H,
                        const double* rMu, const double* rVar, const double* Rt, int64_t lds, int64_t batch, double* lp,
                        double* comp, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    const int wpb = 4;
    const unsigned grid = (unsigned)((batch + wpb - 1) / wpb);
    hipLaunchKernelGGL(k_prior, dim3(grid), dim3(64 * wpb), 0, st, P, birth, death, tH, H, rMu, rVar, Rt, lds, batch, lp, comp);
    return hipGetLastError();
}

}  // namespace mcd
