// k_w_logpdf.hip -- batched log-density of raw x through the multiply form z = W (x - mu) (gfx950); see w_device.hpp.
#include "w_device.hpp"

#include <stdlib.h>

namespace mcd {

#ifdef MCD_W_STAMP
__device__ unsigned long long g_wdbg[128];
#endif

void make_wplan(int Rw, int nw, bool backward, WPlan& pl)
{
    pl = WPlan{};
    pl.nw = nw;
    pl.R = Rw;
    int cnt[16], first[16], base[17];
    base[0] = 0;
    for (int bi = 0; bi < Rw; ++bi) {
        cnt[bi] = backward ? 32 * (Rw - bi) : 32 * (bi + 1);
        first[bi] = backward ? 32 * bi : 0;
        base[bi + 1] = base[bi] + cnt[bi];
    }
    const int total = base[Rw];
    int seg = 0;
    for (int bi = 0; bi <= Rw; ++bi) pl.blk_seg0[bi] = -1;
    for (int w = 0; w < nw; ++w) {
        pl.wave_seg0[w] = seg;
        const int lo = (int)((int64_t)total * w / nw), hi = (int)((int64_t)total * (w + 1) / nw);
        int t = lo;
        while (t < hi) {
            int bi = 0;
            while (base[bi + 1] <= t) ++bi;
            const int e = (hi < base[bi + 1]) ? hi : base[bi + 1];
            pl.seg_bi[seg] = bi;
            pl.seg_p0[seg] = first[bi] + (t - base[bi]);
            pl.seg_p1[seg] = first[bi] + (e - base[bi]);
            pl.seg_off[seg] = t;
            if (pl.blk_seg0[bi] < 0) pl.blk_seg0[bi] = seg;
            ++seg;
            t = e;
        }
    }
    for (int w = nw; w <= 16; ++w) pl.wave_seg0[w] = seg;
    pl.n_seg = seg;
    pl.blk_seg0[Rw] = seg;
    for (int bi = Rw - 1; bi >= 0; --bi)
        if (pl.blk_seg0[bi] < 0) pl.blk_seg0[bi] = pl.blk_seg0[bi + 1];
}

template <int BT, int PD, int MAXT>
__global__ void __launch_bounds__(MAXT) k_w_logpdf(MvnDev M, const WPlan* __restrict__ plan, const double* __restrict__ X, int64_t ldx,
                                                   int64_t batch, double* __restrict__ ll)
{
    extern __shared__ double lds[];
    const WPlan& pl = *plan;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int NPs = 64 * M.Rw;
    double* xs = lds;                                   // [BT][NPs]
    double* part = xs + (size_t)BT * NPs;               // [n_seg][BT][64]
    double* red = part + (size_t)pl.n_seg * BT * 64;    // [nw][BT]
    const int64_t b0 = (int64_t)blockIdx.x * BT;
#ifdef MCD_W_STAMP
    unsigned long long ts[6];
    ts[0] = __builtin_amdgcn_s_memtime();
#define W_T(i) ts[i] = __builtin_amdgcn_s_memtime();
#else
#define W_T(i)
#endif
    // chain data first: vector memory returns in order, so the x loads must not queue behind the factor prefetch
    constexpr int XPT = 4;                              // x elements per thread (host guarantees BT * NPs <= XPT * blockDim)
    double xr[XPT];
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
        const int idx = threadIdx.x + u * blockDim.x;
        xr[u] = 0.0;
        if (idx < BT * NPs) {
            const int c = idx / NPs, j = idx - c * NPs;
            const int64_t b = (b0 + c < batch) ? b0 + c : batch - 1;
            if (j < M.n) xr[u] = X[b * ldx + j] - M.mu[j];             // dxs = xs - mu
        }
    }
    WCursor cu;
    wd2 ring[PD];
    w_prefetch<PD>(M.Wf, pl, wave, lane, cu, ring);
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
        const int idx = threadIdx.x + u * blockDim.x;
        if (idx < BT * NPs) xs[idx] = xr[u];
    }
    W_T(1)
    __syncthreads();
    W_T(2)
    w_pass<BT, PD>(M.Wf, pl, wave, lane, cu, ring, xs, NPs, part);
    W_T(3)
    __syncthreads();
    W_T(4)
    double q[BT];
#pragma unroll
    for (int c = 0; c < BT; ++c) q[c] = 0.0;
    for (int bi = wave; bi < M.Rw; bi += nw) {
        double z[BT];
        w_block_rows<BT>(pl, bi, lane, part, z);
#pragma unroll
        for (int c = 0; c < BT; ++c) q[c] = fma(z[c], z[c], q[c]);
    }
#pragma unroll
    for (int c = 0; c < BT; ++c) {
        const double s = w_wave_sum(q[c]);
        if (lane == 0) red[wave * BT + c] = s;
    }
    __syncthreads();
    if (threadIdx.x < BT && b0 + threadIdx.x < batch) {
        double tot = 0.0;
        for (int w = 0; w < nw; ++w) tot += red[w * BT + threadIdx.x];
        ll[b0 + threadIdx.x] = M.c + (-0.5) * (M.logdet + tot);            // app/Probability.hs:169
    }
#ifdef MCD_W_STAMP
    W_T(5)
    if (blockIdx.x == 7 && lane == 0)
        for (int i = 0; i < 5; ++i) g_wdbg[wave * 8 + i] = ts[i + 1] - ts[i];
#endif
}

static int plan_index(int nw) { return nw == 1 ? 0 : nw == 2 ? 1 : nw == 4 ? 2 : nw == 8 ? 3 : 4; }

template <int BT>
static hipError_t launch_w(const MvnDev& M, int nw, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st)
{
    const int NPs = 64 * M.Rw;
    const int max_seg = nw + M.Rw - 1;
    const size_t sh = sizeof(double) * ((size_t)BT * NPs + (size_t)max_seg * BT * 64 + (size_t)nw * BT);
    const unsigned grid = (unsigned)((batch + BT - 1) / BT);
    static const int env_pd = getenv("MCD_W_PD") ? atoi(getenv("MCD_W_PD")) : 0;
    const int pd = env_pd ? env_pd : 16;
    if (nw > 8 || pd <= 8)
        hipLaunchKernelGGL((k_w_logpdf<BT, 8, 1024>), dim3(grid), dim3(64 * nw), sh, st, M, M.plan_f + plan_index(nw), X, ldx, batch, ll);
    else if (pd <= 16)
        hipLaunchKernelGGL((k_w_logpdf<BT, 16, 512>), dim3(grid), dim3(64 * nw), sh, st, M, M.plan_f + plan_index(nw), X, ldx, batch, ll);
    else if (pd <= 32)
        hipLaunchKernelGGL((k_w_logpdf<BT, 32, 512>), dim3(grid), dim3(64 * nw), sh, st, M, M.plan_f + plan_index(nw), X, ldx, batch, ll);
    else
        hipLaunchKernelGGL((k_w_logpdf<BT, 40, 512>), dim3(grid), dim3(64 * nw), sh, st, M, M.plan_f + plan_index(nw), X, ldx, batch, ll);
    return hipGetLastError();
}

hipError_t launch_logpdf_w(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    static const int env_nw = getenv("MCD_W_NW") ? atoi(getenv("MCD_W_NW")) : 0;
    static const int env_bt = getenv("MCD_W_BT") ? atoi(getenv("MCD_W_BT")) : 0;
    int nw = env_nw ? env_nw : 8;
    int bt = env_bt ? env_bt : (batch <= 256 ? 1 : batch <= 2048 ? 2 : 4);
    if (bt == 1) return launch_w<1>(M, nw, X, ldx, batch, ll, st);
    if (bt == 2) return launch_w<2>(M, nw, X, ldx, batch, ll, st);
    return launch_w<4>(M, nw, X, ldx, batch, ll, st);
}

}  // namespace mcd

#ifdef MCD_W_STAMP
extern "C" int mcd_w_debug_stamps(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mcd::g_wdbg), 128 * sizeof(unsigned long long));
}
#endif
