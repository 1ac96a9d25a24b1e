// k_tree_logpdf.hip -- tree state -> log-likelihood (+ root-branch Jacobian) (gfx950).  Device code: mvn_device.hpp.
#include "mvn_device.hpp"
#include <type_traits>

namespace mcd {

template <int R, int BT, int CW, int LW>
__global__ void __launch_bounds__(64 * (CW + LW)) k_tree_logpdf(MvnDev M, TreeDev T, const double* __restrict__ H,
                                                                const double* __restrict__ Rt, int64_t lds,
                                                                const double* __restrict__ tH,
                                                                const double* __restrict__ rMu, int64_t batch,
                                                                double* __restrict__ ll, double* __restrict__ logjac)
{
    MCD_KERNEL_HEAD
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

template <int R>
static hipError_t launch_tree_logpdf_R(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                                       const double* tH, const double* rMu, int64_t batch, double* ll, double* logjac,
                                       hipStream_t st)
{
    const Geometry g = pick_geometry(batch);
    constexpr int LW = Cfg<R>::LW;
    if (g.cw == 2) {
        const unsigned grid = (unsigned)((batch + 1) / 2);
        hipLaunchKernelGGL((k_tree_logpdf<R, 1, 2, LW>), dim3(grid), dim3(64 * (2 + LW)), 0, st, M, T, H, Rt, lds, tH, rMu, batch, ll, logjac);
    } else if (g.bt == 1) {
        const unsigned grid = (unsigned)((batch + 3) / 4);
        hipLaunchKernelGGL((k_tree_logpdf<R, 1, 4, LW>), dim3(grid), dim3(64 * (4 + LW)), 0, st, M, T, H, Rt, lds, tH, rMu, batch, ll, logjac);
    } else {                                               // large batches: two chains per compute wave share every factor read
        const unsigned grid = (unsigned)((batch + 7) / 8);
        hipLaunchKernelGGL((k_tree_logpdf<R, 2, 4, LW>), dim3(grid), dim3(64 * (4 + LW)), 0, st, M, T, H, Rt, lds, tH, rMu, batch, ll, logjac);
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
#define CALL(R) launch_tree_logpdf_R<R>(M, T, H, Rt, lds, tH, rMu, batch, ll, logjac, st)
    MCD_DISPATCH_R(M.R, CALL)
#undef CALL
}

}  // namespace mcd
