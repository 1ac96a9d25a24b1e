// k_tree_logpdf.hip -- tree state -> log-likelihood (+ root-branch Jacobian) (gfx950).  Device code: mvn_device.hpp.
//
// PRIOR variant (the two-launch Metropolis-Hastings step, mh_capi.cpp): the first J.n_wgs workgroups do not sweep -- each of
// their waves evaluates the ln prior of one chain's proposed state (mh_prior_role.hpp), which depends on the proposal only,
// like the likelihood the other workgroups compute: a lock step then costs accept + propose + max(likelihood, prior) instead of
// their sum.  The plain variant is what every other caller launches; its code is unchanged.
#define MCD_BID (blockIdx.x - bid_off)
#include "mvn_device.hpp"
#include "mh_prior_role.hpp"
#include <type_traits>

namespace mcd {

struct MhPriorJob {
    MhDev M;
    PriorDev P;
    int n_wgs;
    int wpc;       // waves per chain: 1, or 2 (mh_prior_role2)
};
struct NoJob {};

template <int R, int BT, int CW, int LW, bool PRIOR>
__global__ void __launch_bounds__(64 * (CW + LW)) k_tree_logpdf(MvnDev M, TreeDev T, const double* __restrict__ H,
                                                                const double* __restrict__ Rt, int64_t lds,
                                                                const double* __restrict__ tH,
                                                                const double* __restrict__ rMu, int64_t batch,
                                                                double* __restrict__ ll, double* __restrict__ logjac,
                                                                std::conditional_t<PRIOR, MhPriorJob, NoJob> J)
{
    unsigned bid_off = 0;
    if constexpr (PRIOR) bid_off = (unsigned)J.n_wgs;
    MCD_KERNEL_HEAD
    if constexpr (PRIOR) {
        if (blockIdx.x < bid_off) {                        // the prior role: a chain per wave (or per two), its state in a slice of the ring
            if (J.wpc == 2) {
                const int64_t pb = (int64_t)blockIdx.x * ((CW + LW) / 2) + (wave >> 1);
                const bool valid = pb < J.M.batch;
                mh_prior_role2(J.M, J.P, valid ? pb : J.M.batch - 1, valid, wave & 1, lane,
                               reinterpret_cast<double*>(ring) + (size_t)(wave >> 1) * mh_prior_role2_doubles(J.M.n_nodes));
                return;
            }
            const int64_t pb = (int64_t)blockIdx.x * (CW + LW) + wave;
            if (pb >= J.M.batch) return;
            double* hs = reinterpret_cast<double*>(ring) + (size_t)wave * 2 * J.M.n_nodes;
            mh_prior_role(J.M, J.P, pb, lane, hs, hs + J.M.n_nodes);
            return;
        }
    }
    MCD_ACC_DECL
    if (wave >= CW) {                                      // loader role
        Stage<R, LW> st;
        const int lw = wave - CW;
        fwd_loader_prologue<R, LW>(M.Ft, ring, st, lw, lane);
        lds_barrier();
        fwd_loader_start<R, LW>(M.Ft, st, lw, lane);
        fwd_loader<R, LW, 0>(M.Ft, ring, st, lw, lane, ncols MCD_ACC_ARGS);
        return;
    }
    double d[R][BT], dist[R][BT];
    // a sampler's batch (two compute waves per workgroup) on trees up to 258 nodes: the state rows go through LDS (10 KiB beside
    // the ring: two workgroups still fit a CU); elsewhere the gather from global memory
    constexpr bool STAGED = (R <= 4 && BT == 1 && CW == 2);
    __shared__ double tstage[STAGED ? CW * 2 * (64 * R + 64) : 1];
    if constexpr (STAGED)
        load_tree_staged<R>(d, dist, M, T, H, Rt, lds, tH, rMu, b0, batch, lane, tstage + (size_t)wave * 2 * (64 * R + 64));
    else
        load_tree<R, BT>(d, dist, M, T, H, Rt, lds, tH, rMu, b0, batch, lane);
    if (logjac != nullptr && lane == 0) {
#pragma unroll
        for (int c = 0; c < BT; ++c)
            if (b0 + c < batch) logjac[b0 + c] = log(1.0 / dist[0][c]);  // app/Probability.hs:394, 409
    }
    lds_barrier();
    fwd_compute<R, BT, 0>(d, ring, lane, ncols MCD_ACC_ARGS);
    finish_ll<R, BT>(d, M, b0, batch, ll, lane);
}

template <int R, bool PRIOR, class JOB>
static hipError_t launch_tree_logpdf_R(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                                       const double* tH, const double* rMu, int64_t batch, double* ll, double* logjac, const JOB& job,
                                       hipStream_t st)
{
    const Geometry g = pick_geometry(batch);
    constexpr int LW = Cfg<R>::LW;
    JOB J = job;
    auto prior_wgs = [&](int waves) {                      // workgroups of the prior role in front of the sweeping ones
        if constexpr (PRIOR) {
            // two waves per chain while every workgroup of both roles is resident at once (two of these workgroups fit a CU)
            const int per_wg = waves / 2;
            const int bt_eff = (R >= 16) ? 1 : g.bt;
            const int64_t like_wgs = (batch + g.cw * bt_eff - 1) / (g.cw * bt_eff);
            const bool two = (waves % 2 == 0) && like_wgs + (J.M.batch + per_wg - 1) / per_wg <= 512 &&
                             (size_t)per_wg * mh_prior_role2_doubles(J.M.n_nodes) * sizeof(double) <= (size_t)2 * Cfg<R>::SU * 64 * 16;
            J.wpc = two ? 2 : 1;
            const int chains = two ? per_wg : waves;
            J.n_wgs = (int)((J.M.batch + chains - 1) / chains);
            return (unsigned)J.n_wgs;
        } else {
            return 0u;
        }
    };
    if (g.cw == 2) {
        const unsigned grid = (unsigned)((batch + 1) / 2) + prior_wgs(2 + LW);
        hipLaunchKernelGGL((k_tree_logpdf<R, 1, 2, LW, PRIOR>), dim3(grid), dim3(64 * (2 + LW)), 0, st, M, T, H, Rt, lds, tH, rMu, batch, ll, logjac, J);
    } else if (g.bt == 1) {
        const unsigned grid = (unsigned)((batch + 3) / 4) + prior_wgs(4 + LW);
        hipLaunchKernelGGL((k_tree_logpdf<R, 1, 4, LW, PRIOR>), dim3(grid), dim3(64 * (4 + LW)), 0, st, M, T, H, Rt, lds, tH, rMu, batch, ll, logjac, J);
    } else if constexpr (R >= 16) {
        // (two chains per compute wave do not fit the register file at R = 16: 1 188 spilled registers; such a batch -- more than 4096 chains on
        // the sweep -- is only reached with the form forced, the automatic choice takes the multiply form there)
        const unsigned grid = (unsigned)((batch + 3) / 4) + prior_wgs(4 + LW);
        hipLaunchKernelGGL((k_tree_logpdf<R, 1, 4, LW, PRIOR>), dim3(grid), dim3(64 * (4 + LW)), 0, st, M, T, H, Rt, lds, tH, rMu, batch, ll, logjac, J);
    } else {                                               // large batches: two chains per compute wave share every factor read
        const unsigned grid = (unsigned)((batch + 7) / 8) + prior_wgs(4 + LW);
        hipLaunchKernelGGL((k_tree_logpdf<R, 2, 4, LW, PRIOR>), dim3(grid), dim3(64 * (4 + LW)), 0, st, M, T, H, Rt, lds, tH, rMu, batch, ll, logjac, J);
    }
    return hipGetLastError();
}

// Each kernel file is compiled four times (-DMCD_RGROUP=0: R in {1,2,3,4}; 1: {6,8}; 2: {12}; 3: {16}) so that
// the template instantiations build in parallel and the big ones never share a translation unit.
#ifndef MCD_RGROUP
#define MCD_RGROUP 0
#endif
#if MCD_RGROUP == 0
#define MCD_DISPATCH_R(R_, CALL) \
    switch (R_) { case 1: return CALL(1); case 2: return CALL(2); case 3: return CALL(3); case 4: return CALL(4); default: return hipErrorInvalidValue; }
#elif MCD_RGROUP == 1
#define MCD_DISPATCH_R(R_, CALL) \
    switch (R_) { case 6: return CALL(6); case 8: return CALL(8); default: return hipErrorInvalidValue; }
#elif MCD_RGROUP == 2
#define MCD_DISPATCH_R(R_, CALL) \
    switch (R_) { case 12: return CALL(12); default: return hipErrorInvalidValue; }
#else
#define MCD_DISPATCH_R(R_, CALL) \
    switch (R_) { case 16: return CALL(16); default: return hipErrorInvalidValue; }
#endif
#define MCD_CAT2(a, b) a##b
#define MCD_CAT(a, b) MCD_CAT2(a, b)

#if MCD_RGROUP == 0
hipError_t launch_tree_logpdf_g1(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                              const double* tH, const double* rMu, int64_t batch, double* ll, double* logjac,
                              hipStream_t st);
hipError_t launch_tree_logpdf_g2(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                              const double* tH, const double* rMu, int64_t batch, double* ll, double* logjac,
                              hipStream_t st);
hipError_t launch_tree_logpdf_g3(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                              const double* tH, const double* rMu, int64_t batch, double* ll, double* logjac,
                              hipStream_t st);
hipError_t launch_tree_logpdf(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                              const double* tH, const double* rMu, int64_t batch, double* ll, double* logjac,
                              hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (use_split(M, batch)) return launch_tree_logpdf_split(M, T, H, Rt, lds, tH, rMu, batch, ll, logjac, st);
    if (use_wide(M, batch)) return launch_tree_logpdf_wide(M, T, H, Rt, lds, tH, rMu, batch, ll, logjac, st);
    if (M.R == 6 || M.R == 8) return launch_tree_logpdf_g1(M, T, H, Rt, lds, tH, rMu, batch, ll, logjac, st);
    if (M.R == 12) return launch_tree_logpdf_g2(M, T, H, Rt, lds, tH, rMu, batch, ll, logjac, st);
    if (M.R == 16) return launch_tree_logpdf_g3(M, T, H, Rt, lds, tH, rMu, batch, ll, logjac, st);
#else
hipError_t MCD_CAT(launch_tree_logpdf_g, MCD_RGROUP)(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                              const double* tH, const double* rMu, int64_t batch, double* ll, double* logjac,
                              hipStream_t st)
{
#endif
#define CALL(R) launch_tree_logpdf_R<R, false, NoJob>(M, T, H, Rt, lds, tH, rMu, batch, ll, logjac, NoJob{}, st)
    MCD_DISPATCH_R(M.R, CALL)
#undef CALL
}

// ---- the same launch with the prior role in front (Metropolis-Hastings, two-launch path) ----
#if MCD_RGROUP == 0
hipError_t launch_tree_logpdf_prior_g1(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                       const double* rMu, int64_t batch, double* ll, double* logjac, const MhDev& J, const PriorDev& JP, hipStream_t st);
hipError_t launch_tree_logpdf_prior_g2(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                       const double* rMu, int64_t batch, double* ll, double* logjac, const MhDev& J, const PriorDev& JP, hipStream_t st);
hipError_t launch_tree_logpdf_prior_g3(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                       const double* rMu, int64_t batch, double* ll, double* logjac, const MhDev& J, const PriorDev& JP, hipStream_t st);
// the sweep serves this launch, and the slices of the ring the prior waves use fit it
bool tree_logpdf_can_carry_prior(const MvnDev& M, int64_t batch, int n_nodes)
{
    if (batch <= 0 || use_split(M, batch) || use_wide(M, batch)) return false;
    const Geometry g = pick_geometry(batch);
    const int lw = (M.R >= 12) ? 4 : 2, su = (M.R >= 12) ? 64 : 32;
    return (size_t)(g.cw + lw) * 2 * (size_t)n_nodes * sizeof(double) <= (size_t)2 * su * 64 * 16;
}
hipError_t launch_tree_logpdf_with_prior(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                         const double* rMu, int64_t batch, double* ll, double* logjac, const MhDev& J, const PriorDev& JP,
                                         hipStream_t st)
{
    if (!tree_logpdf_can_carry_prior(M, batch, J.n_nodes) || J.batch != batch) return hipErrorInvalidValue;
    if (M.R == 6 || M.R == 8) return launch_tree_logpdf_prior_g1(M, T, H, Rt, lds, tH, rMu, batch, ll, logjac, J, JP, st);
    if (M.R == 12) return launch_tree_logpdf_prior_g2(M, T, H, Rt, lds, tH, rMu, batch, ll, logjac, J, JP, st);
    if (M.R == 16) return launch_tree_logpdf_prior_g3(M, T, H, Rt, lds, tH, rMu, batch, ll, logjac, J, JP, st);
#else
hipError_t MCD_CAT(launch_tree_logpdf_prior_g, MCD_RGROUP)(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                                       const double* tH, const double* rMu, int64_t batch, double* ll, double* logjac, const MhDev& J,
                                       const PriorDev& JP, hipStream_t st)
{
#endif
    const MhPriorJob job{J, JP, 0, 1};
#define CALL(R) launch_tree_logpdf_R<R, true, MhPriorJob>(M, T, H, Rt, lds, tH, rMu, batch, ll, logjac, job, st)
    MCD_DISPATCH_R(M.R, CALL)
#undef CALL
}

}  // namespace mcd
