// k_mh_inc.hip -- the two-launch Metropolis-Hastings path on large trees (more than 320 nodes: k_mh_step_wg + the row-split
// likelihood) without the likelihood launch on most steps (gfx950; round 3).
//
// 70 % of the reference's proposal cycle (app/Definitions.hs:127-278) moves at most a handful of branch distances: slide a node,
// scale one branch rate, sub-tree moves of small sub trees.  The streaming chain kernel (k_mh_chain_big.hip) keeps z = L^-1 (d - mu)
// of every chain's current state in registers and evaluates such a proposal as z' = z + sum_j delta_j W[:, j], q' = |z'|^2; here the
// same algebra runs beside the per-step launches, with z in global memory:
//   k_mh_step_wg     (k_mh.hip) evaluates the sparse proposal it has just made itself (mh_inc_device.hpp: mh_inc_ll_block): delta =
//                    X1 - X0 (proposed and current distances), the moved rows found in row order, their 8-KiB columns of W = L^-1
//                    (MvnDev::Wc) read coalesced by the workgroup's 256 threads, z' stored for the accept, ll' written where the
//                    row-split kernel would have written it -- no likelihood launch on such a step.  A proposal that cannot move the
//                    likelihood (birth rate, death rate, rate variance) gets ll' = ll.  (As a launch of its own the same code took
//                    35.4 us per lock step at 1025 nodes x 512 chains; fused: see DESIGN.md.)
//   dense proposals  the row-split kernel in its z-writing mode (launch_logpdf_split_z): ll' and z' tile-major in its scratch.
//   k_mh_step_wg     on accept copies X1 -> X0 and z' -> zcur from whichever place the pending proposal left it (MhInc::mode).
//   k_mh_inc_init / k_mh_inc_take_z   at the start of a run and every 256 steps: X0 from the states, zcur from a full product.
// ln likelihood values agree with a full evaluation to rounding, not bit for bit (decisions, states, the other posterior terms
// stay the same bits); MCD_MH_INCREMENTAL=0 keeps the full evaluation at every step.
// Reference of the quantity: logDensityFullMultivariateNormal, app/Probability.hs:166-173; distances :195-207.
#include "mh_inc_device.hpp"

namespace mcd {

constexpr int kIncT = 256;

// X0 = the distances of the current states (the arithmetic of k_mh_step_wg's X1 and of load_tree)
__global__ __launch_bounds__(kIncT) void k_mh_inc_init(MhDev M, TreeDev T, MhInc I, int n_dim, int64_t ldx)
{
    const int64_t b = blockIdx.x, B = M.batch;
    const double* H = M.H + b * M.ld;
    const double* R = M.R + b * M.ld;
    const double s = M.sc[2 * B + b] * M.sc[3 * B + b];
    const int rr = T.root_right;
    for (int j = threadIdx.x; j < n_dim; j += kIncT) {
        const int a = T.slot_node[j], pa = T.slot_parent[j];
        double d = (H[pa] - H[a]) * R[a];
        if (j == 0) d = d + (H[0] - H[rr]) * R[rr];
        d = d * s;
        I.X0[b * ldx + j] = d;
    }
}

// dst[b0 + c] <- the z tiles of a full product on `count` chains starting at chain b0 (tile-major: the row-split kernel's scratch); dst = zcur
// (a full product on X0) or zprop (a dense proposal of a batch beyond the row-split kernel's 1024 chains, taken chunk by chunk)
__global__ __launch_bounds__(kIncT) void k_mh_inc_take_z(MhInc I, double* __restrict__ dst, int64_t b0)
{
    const int64_t c = blockIdx.x;                            // chain within the chunk
    const double* zt = I.zt + ((c >> 4) * I.nr) * 16 + (c & 15);
    double* zc = dst + (b0 + c) * I.NPz;
    for (int i = threadIdx.x; i < I.NPz; i += kIncT) zc[i] = (i < I.nr) ? zt[(int64_t)i * 16] : 0.0;
}

hipError_t launch_mh_inc_init(const MhDev& M, const TreeDev& T, const MhInc& I, int n_dim, int64_t ldx, hipStream_t st)
{
    hipLaunchKernelGGL(k_mh_inc_init, dim3((unsigned)M.batch), dim3(kIncT), 0, st, M, T, I, n_dim, ldx);
    return hipGetLastError();
}

hipError_t launch_mh_inc_take_z(const MhInc& I, double* dst, int64_t b0, int64_t count, hipStream_t st)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_mh_inc_take_z, dim3((unsigned)count), dim3(kIncT), 0, st, I, dst, b0);
    return hipGetLastError();
}

}  // namespace mcd
