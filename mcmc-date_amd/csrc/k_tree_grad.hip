// k_tree_grad.hip -- tree state -> log-likelihood + gradient wrt the state (gfx950).  Device code: mvn_device.hpp.
#include "mvn_device.hpp"
#include <type_traits>

namespace mcd {

// Chain rule from g = d ll / d distances back to heights, rates, tH, rMu (SURVEY.md 8a A7:
// d ll/d r_v = s g t_v, d ll/d h_v = s (sum_children g_c r_c - g_v r_v), d ll/d tH = g.d / tH).  One chain per compute wave;
// e[v] = s * g[row(v)] * rate[v] is exchanged through the (by then idle) LDS ring, one private
// region per wave.
template <int R, int CW, int LW>
__global__ void __launch_bounds__(64 * (CW + LW)) k_tree_grad(MvnDev M, TreeDev T, const double* __restrict__ H,
                                                              const double* __restrict__ Rt, int64_t lds,
                                                              const double* __restrict__ tH,
                                                              const double* __restrict__ rMu, int64_t batch,
                                                              double* __restrict__ ll, double* __restrict__ gH,
                                                              double* __restrict__ gR, double* __restrict__ gtH,
                                                              double* __restrict__ grMu)
{
    constexpr int BT = 1;
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
    double d[R][1], dist[R][1];
    load_tree<R, 1>(d, dist, M, T, H, Rt, lds, tH, rMu, b0, batch, lane);
    lds_barrier();
    fwd_compute<R, 1, 0>(d, ring, lane, ncols MCD_ACC_ARGS);
    finish_ll<R, 1>(d, M, b0, batch, ll, lane);
#pragma unroll
    for (int k = 0; k < R; ++k) d[k][0] *= M.invdiag[64 * k + lane];
    {
        bool started = false;
        bwd_compute<R, 1, R - 1>(d, ring, lane, ncols, started);
    }
    // now d = y = Sigma^-1 (dist - mu); g = -y.  The last barrier of the sweep has passed: the ring
    // is free.  n_nodes_pad <= 64 R + 64 doubles per compute wave fit in it (CW * 8.5 KiB <= 64 KiB).
    if (b0 >= batch) return;               // no barriers below
    const int64_t b = b0;
    double* e = reinterpret_cast<double*>(ring) + (size_t)wave * T.n_nodes_pad;
    const double s = tH[b] * rMu[b];
    const double* h = H + b * lds;
    const double* r = Rt + b * lds;
    double gd = 0.0;
    if (lane == 0) e[0] = 0.0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int row = 64 * k + lane;
        const int a = T.slot_node[row];
        const double g = -d[k][0];
        gd = fma(g, dist[k][0], gd);
        if (a >= 0) {
            const int pa = T.slot_parent[row];
            const double sg = s * g;
            gR[b * lds + a] = sg * (h[pa] - h[a]);
            e[a] = sg * r[a];
            if (row == 0) {
                const int a2 = T.root_right;
                gR[b * lds + a2] = sg * (h[0] - h[a2]);
                e[a2] = sg * r[a2];
            }
        }
    }
    const double gdot = wave_sum(gd);
    if (lane == 0) {
        gR[b * lds] = 0.0;                 // stem rate: unused by the likelihood
        gtH[b] = gdot / tH[b];
        grMu[b] = gdot / rMu[b];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's e[] writes have landed (wave-private region)
    for (int v = lane; v < T.n_nodes; v += 64) {
        double acc = (v == 0) ? 0.0 : -e[v];
        for (int ci = T.child_ptr[v]; ci < T.child_ptr[v + 1]; ++ci) acc += e[T.child_idx[ci]];
        gH[b * lds + v] = acc;
    }
}

template <int R>
static hipError_t launch_tree_grad_R(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                                     const double* tH, const double* rMu, int64_t batch, double* ll, double* gH,
                                     double* gR, double* gtH, double* grMu, hipStream_t st)
{
    // (as k_grad.hip: from R = 8 two compute waves per workgroup, from R = 12 two loader waves -- one wave per SIMD, no spilled registers)
    auto go = [&](auto cw_tag) {
        constexpr int CW = decltype(cw_tag)::value, LW = (R == 12) ? 2 : Cfg<R>::LW;
        const unsigned grid = (unsigned)((batch + CW - 1) / CW);
        hipLaunchKernelGGL((k_tree_grad<R, CW, LW>), dim3(grid), dim3(64 * (CW + LW)), 0, st, M, T, H, Rt, lds, tH, rMu, batch,
                       ll, gH, gR, gtH, grMu);
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
hipError_t launch_tree_grad_g1(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                            const double* tH, const double* rMu, int64_t batch, double* ll, double* gH, double* gR,
                            double* gtH, double* grMu, hipStream_t st);
hipError_t launch_tree_grad_g2(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                            const double* tH, const double* rMu, int64_t batch, double* ll, double* gH, double* gR,
                            double* gtH, double* grMu, hipStream_t st);
hipError_t launch_tree_grad_g3(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                            const double* tH, const double* rMu, int64_t batch, double* ll, double* gH, double* gR,
                            double* gtH, double* grMu, hipStream_t st);
hipError_t launch_tree_grad(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                            const double* tH, const double* rMu, int64_t batch, double* ll, double* gH, double* gR,
                            double* gtH, double* grMu, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (use_split_grad(M, batch) && (const double*)gH != Rt && (const double*)gR != H)   // (either output may be its own input, not the other's)
        return launch_tree_grad_split(M, T, H, Rt, lds, tH, rMu, batch, ll, gH, gR, gtH, grMu, st);
    if (use_wide_grad(M, batch)) {
        if (M.n <= 256) return launch_tree_grad_wide(M, T, H, Rt, lds, tH, rMu, batch, ll, gH, gR, gtH, grMu, st);
        if ((const double*)gH != H && (const double*)gH != Rt)      // (the height-gradient rows double as scratch above 256)
            return launch_tree_grad_wide_mc(M, T, H, Rt, lds, tH, rMu, batch, ll, gH, gR, gtH, grMu, st);
    }
    if (M.R == 6 || M.R == 8) return launch_tree_grad_g1(M, T, H, Rt, lds, tH, rMu, batch, ll, gH, gR, gtH, grMu, st);
    if (M.R == 12) return launch_tree_grad_g2(M, T, H, Rt, lds, tH, rMu, batch, ll, gH, gR, gtH, grMu, st);
    if (M.R == 16) {
        // N > 768: no sweep form of the tree gradient (k_grad.hip has the reason): the row split in pieces of at most 1024 chains, whatever
        // the batch and the form asked for.  Its one restriction: an output that aliases the OTHER input array (the height gradient over the
        // rates or the reverse) is refused -- include/mcmcdate_mvn.h says so.
        if (M.split == nullptr || (const double*)gH == Rt || (const double*)gR == H) return hipErrorInvalidValue;
        for (int64_t c0 = 0; c0 < batch; c0 += kSplitMaxBatch) {
            const int64_t cnt = (batch - c0 < kSplitMaxBatch) ? batch - c0 : kSplitMaxBatch;
            if (hipError_t e = launch_tree_grad_split(M, T, H + c0 * lds, Rt + c0 * lds, lds, tH + c0, rMu + c0, cnt, ll + c0, gH + c0 * lds, gR + c0 * lds,
                                                      gtH + c0, grMu + c0, st))
                return e;
        }
        return hipSuccess;
    }
#else
hipError_t MCD_CAT(launch_tree_grad_g, MCD_RGROUP)(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                            const double* tH, const double* rMu, int64_t batch, double* ll, double* gH, double* gR,
                            double* gtH, double* grMu, hipStream_t st)
{
#endif
#if MCD_RGROUP == 3
    return hipErrorInvalidValue;                           // (R = 16: launch_tree_grad takes the row split, see there)
#else
#define CALL(R) launch_tree_grad_R<R>(M, T, H, Rt, lds, tH, rMu, batch, ll, gH, gR, gtH, grMu, st)
    MCD_DISPATCH_R(M.R, CALL)
#undef CALL
#endif
}

}  // namespace mcd
