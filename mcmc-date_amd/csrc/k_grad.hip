// k_grad.hip -- log-density + gradient wrt x (gfx950).  Device code: mvn_device.hpp.
#include "mvn_device.hpp"
#include <type_traits>

namespace mcd {

template <int R, int BT, int CW, int LW>
__global__ void __launch_bounds__(64 * (CW + LW)) k_grad(MvnDev M, const double* __restrict__ X, int64_t ldx,
                                                         int64_t batch, double* __restrict__ ll, double* __restrict__ G,
                                                         int64_t ldg)
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
        bool started = false;
        bwd_loader<R, LW, R - 1>(M.Ut, ring, st, lw, lane, ncols, started);
        return;
    }
    double d[R][BT];
    load_rawx<R, BT>(d, M, X, ldx, b0, batch, lane);
    lds_barrier();
    fwd_compute<R, BT, 0>(d, ring, lane, ncols MCD_ACC_ARGS);
    finish_ll<R, BT>(d, M, b0, batch, ll, lane);
    // backward: y = L^-T z.  Row scaling first (z_r / L_rr), then the mirrored sweep.
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const double iv = M.invdiag[64 * k + lane];
#pragma unroll
        for (int c = 0; c < BT; ++c) d[k][c] *= iv;
    }
    bool started = false;
    bwd_compute<R, BT, R - 1>(d, ring, lane, ncols, started);
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int row = 64 * k + lane;
#pragma unroll
        for (int c = 0; c < BT; ++c)
            if (row < M.n && b0 + c < batch) G[(b0 + c) * ldg + row] = -d[k][c];
    }
}

template <int R>
static hipError_t launch_grad_R(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, double* G,
                                int64_t ldg, hipStream_t st)
{
    // (the gradient holds the forward AND the backward sweep's staging: above R = 6 four compute waves per workgroup, and at R = 12 four
    // loader waves, no longer fit two waves per SIMD -- 140 .. 1 219 spilled registers -- so those sizes take two compute waves, and two
    // loaders from R = 12: one wave per SIMD, the whole register file.  The sweeps are the fallback there: up to 1024 chains the row split
    // serves the gradient, from 2048 the multiply form.)
    auto go = [&](auto cw_tag) {
        constexpr int CW = decltype(cw_tag)::value, LW = (R == 12) ? 2 : Cfg<R>::LW;
        const unsigned grid = (unsigned)((batch + CW - 1) / CW);
        hipLaunchKernelGGL((k_grad<R, 1, CW, LW>), dim3(grid), dim3(64 * (CW + LW)), 0, st, M, X, ldx, batch, ll, G, ldg);
    };
    if constexpr (R >= 8) {
        go(std::integral_constant<int, 2>{});
    } else {
        if (pick_geometry(batch).cw == 2)
            go(std::integral_constant<int, 2>{});
        else
            go(std::integral_constant<int, 4>{});
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
hipError_t launch_grad_g1(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, double* G, int64_t ldg,
                       hipStream_t st);
hipError_t launch_grad_g2(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, double* G, int64_t ldg,
                       hipStream_t st);
hipError_t launch_grad_g3(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, double* G, int64_t ldg,
                       hipStream_t st);
hipError_t launch_grad(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, double* G, int64_t ldg,
                       hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (use_split_grad(M, batch)) return launch_grad_split(M, X, ldx, batch, ll, G, ldg, st);
    if (use_wide_grad(M, batch)) {
        if (M.n <= 256) return launch_grad_wide(M, X, ldx, batch, ll, G, ldg, st);
        // above 256 the gradient rows double as scratch for z: an in-place call (G == X) keeps the sweep, which reads a chain's
        // x completely before it writes
        if ((const double*)G != X) return launch_grad_wide_mc(M, X, ldx, batch, ll, G, ldg, st);
    }
    if (M.R == 6 || M.R == 8) return launch_grad_g1(M, X, ldx, batch, ll, G, ldg, st);
    if (M.R == 12) return launch_grad_g2(M, X, ldx, batch, ll, G, ldg, st);
    if (M.R == 16) {
        // N > 768: no sweep form of the gradient (16 row blocks of both sweeps' staging do not fit two waves per SIMD: 260 .. 1 200 spilled
        // registers; it was the fallback only -- 207 us at N = 1024 x 512 chains against the row split's 42): whatever the batch and
        // the form asked for, the row split in pieces of at most 1024 chains (in place is fine: its first pass has read every x)
        if (M.split == nullptr) return hipErrorInvalidValue;
        for (int64_t c0 = 0; c0 < batch; c0 += kSplitMaxBatch) {
            const int64_t cnt = (batch - c0 < kSplitMaxBatch) ? batch - c0 : kSplitMaxBatch;
            if (hipError_t e = launch_grad_split(M, X + c0 * ldx, ldx, cnt, ll + c0, G + c0 * ldg, ldg, st)) return e;
        }
        return hipSuccess;
    }
#else
hipError_t MCD_CAT(launch_grad_g, MCD_RGROUP)(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, double* G, int64_t ldg,
                       hipStream_t st)
{
#endif
#if MCD_RGROUP == 3
    return hipErrorInvalidValue;                           // (R = 16: launch_grad takes the row split, see there)
#else
#define CALL(R) launch_grad_R<R>(M, X, ldx, batch, ll, G, ldg, st)
    MCD_DISPATCH_R(M.R, CALL)
#undef CALL
#endif
}

}  // namespace mcd
