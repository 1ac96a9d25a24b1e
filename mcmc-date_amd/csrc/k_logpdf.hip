// k_logpdf.hip -- batched log-density, raw x (gfx950).  Device code: mvn_device.hpp.
#include "mvn_device.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <atomic>

namespace mcd {

template <int R, int BT, int CW, int LW>
__global__ void __launch_bounds__(64 * (CW + LW)) k_logpdf(MvnDev M, const double* __restrict__ X, int64_t ldx,
                                                           int64_t batch, double* __restrict__ ll)
{
    MCD_KERNEL_HEAD
    MCD_ACC_DECL
    MCD_T(0);
    if (wave >= CW) {                                      // loader role
        Stage<R, LW> st;
        const int lw = wave - CW;
        fwd_loader_prologue<R, LW>(M.Ft, ring, st, lw, lane);
        MCD_T(1);
        lds_barrier();
        MCD_T(2);
        fwd_loader_start<R, LW>(M.Ft, st, lw, lane);
        fwd_loader<R, LW, 0>(M.Ft, ring, st, lw, lane, ncols MCD_ACC_ARGS);
        MCD_T(3);
        MCD_ACC_FLUSH(5);
        return;
    }
    double d[R][BT];
    load_rawx<R, BT>(d, M, X, ldx, b0, batch, lane);
    MCD_T(1);
    lds_barrier();
    MCD_T(2);
    fwd_compute<R, BT, 0>(d, ring, lane, ncols MCD_ACC_ARGS);
    MCD_T(3);
    finish_ll<R, BT>(d, M, b0, batch, ll, lane);
    MCD_T(4);
    MCD_ACC_FLUSH(5);
}

template <int R, int BT, int CW, int LW>
static void launch_geom(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st)
{
    const int64_t per_wg = (int64_t)CW * BT;
    const unsigned grid = (unsigned)((batch + per_wg - 1) / per_wg);
    hipLaunchKernelGGL((k_logpdf<R, BT, CW, LW>), dim3(grid), dim3(64 * (CW + LW)), 0, st, M, X, ldx, batch, ll);
}

template <int R>
static hipError_t launch_logpdf_R(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st)
{
    Geometry g = pick_geometry(batch);
    if constexpr (R == 3 || R == 4) {
        // Four loader waves beside the two compute waves at 129 .. 256 dimensions (round 4): one wave pulls 13 - 25 B/clk out of the L2, two do not
        // reach the CU's ingest rate -- N = 256 x 512 chains 7.62 -> 7.41 us per launch, N = 192 5.83 -> 5.50 (same box; 3 waves 8.46 / 6 waves 8.20
        // against 8.82 / 7.42 for 2 / 4 on a slower box); no gain at one or two row blocks.  mcd_set_option "MCD_LOADERS" = 2 keeps two (A/B).
        if (g.cw == 2 && opt_or(OPT_LOADERS, 4) == 4) {
            launch_geom<R, 1, 2, 4>(M, X, ldx, batch, ll, st);
            return hipGetLastError();
        }
    }
    if (g.cw == 2)
        launch_geom<R, 1, 2, Cfg<R>::LW>(M, X, ldx, batch, ll, st);
    else if (g.bt == 1 || R >= 16)                         // (two chains per compute wave do not fit the register file at R = 16: 1 048 spilled registers)
        launch_geom<R, 1, 4, Cfg<R>::LW>(M, X, ldx, batch, ll, st);
    else if constexpr (R < 16)
        launch_geom<R, 2, 4, Cfg<R>::LW>(M, X, ldx, batch, ll, st);
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
int sweep_chunk_columns(int R)
{
    // columns per chunk (Cfg<R>::CCOLS): the swept column count is rounded up to it (extra columns
    // are zero padding).
    return 2 * ((R == 1) ? 32 : (R == 2) ? 16 : (R <= 4) ? 8 : 4);
}

int wide_chain_tiles(int64_t batch)
{
    const int force = opt_or(OPT_WIDE_CT, 0);
    if (force == 1 || force == 2 || force == 4) return force;
    // 16 chains per workgroup while that leaves every CU at most one workgroup; 32 above (two workgroups then share a CU:
    // one stages while the other multiplies).  64 (MCD_WIDE_CT=4) measured slower at every size: its LDS chunk fills the CU.
    return batch <= 256 * 16 ? 1 : 2;
}

static std::atomic<int> g_form{getenv("MCD_WIDE") ? (atoi(getenv("MCD_WIDE")) ? 2 : 1) : 0};   // MCD_FORM_*: the process default

int set_logpdf_form(int form) { return g_form.exchange(form); }

// the form in force for a handle: its own choice (mcd_mvn_set_form) or, if it has none, the process default
int effective_form(const MvnDev& M)
{
    const int own = M.form ? __atomic_load_n(M.form, __ATOMIC_RELAXED) : 0;
    return own != 0 ? own : g_form.load(std::memory_order_relaxed);
}

bool use_wide(const MvnDev& M, int64_t batch)
{
    const int form = effective_form(M);
    if (M.Wt == nullptr || form == 1) return false;
    if (form == 2) return true;
    // measured crossovers (tools/bench_forms.py, profiles/r01_form_crossover.jsonl): at 1024 chains the sweep still wins
    // or ties for every N, at 2048 the multiply form wins from N = 127 up; small N only pays at 8192 chains
    return (M.n >= 96 && batch >= 2048) || (M.n >= 32 && batch >= 8192);
}

bool use_split(const MvnDev& M, int64_t batch)
{
    // measured window (tools/gpu/window.sh, window2.sh; profiles/r02_split_window.jsonl, r02_split_window_240.jsonl; raw x and
    // tree states alike): above N = 256 -- five or more 64-row blocks in the sweep's dependent chain -- the row split wins for every
    // batch from 1 to 1024 chains, by 1.6x at N = 384 to 6.5x at N = 1024; at 240 < N <= 256 up to 128 chains (6.95-7.25 against
    // 7.4-7.5 us); from 256 chains the sweep's single launch-to-result path is shorter (7.7 against 7.9 us at 512 chains: the split
    // pays about two memory round trips for handing the partial sums over); at N = 200 and 224 (13 / 14 row blocks over 8 groups:
    // uneven) and below the sweep wins everywhere; from 2048 chains k_wide takes over
    const int force = opt_or(OPT_SPLIT, -1);               // tests and tuning (mcd_set_option "MCD_SPLIT"): 1 = wherever possible, 0 = never
    if (effective_form(M) != 0 || M.split == nullptr || batch < 1 || batch > kSplitMaxBatch || force == 0) return false;
    if (force == 1) return true;
    if (M.n > 256) return true;
    return M.n > 240 && batch <= 128;
}

bool use_split_grad(const MvnDev& M, int64_t batch)
{
    // the gradient on the row-split schedule: two (tree states: three) launches over 8 row groups x batch / 16 workgroups, where
    // the sweep walks a chain of N / 64 dependent blocks twice and k_wide_grad_mc fills batch / 16 CUs.  MCD_SPLIT as above.
    const int force = opt_or(OPT_SPLIT, -1);
    if (effective_form(M) != 0 || M.split == nullptr || batch < 1 || batch > kSplitMaxBatch || force == 0) return false;
    if (force == 1) return true;
    return M.n > 256 || (M.n > 240 && batch <= 512);      // (measured: tools/gpu/grad_prof.sh; at N = 224 the sweeps are level or ahead)
}

bool use_wide_grad(const MvnDev& M, int64_t batch)
{
    // N <= 256: z and y stay in one LDS chunk (k_wide_grad.hip); above they pass through the output buffer (k_wide_grad_mc.hip)
    return M.Wtb != nullptr && use_wide(M, batch);
}

int padded_blocks(int n)
{
    const int r = (n + 63) / 64;
    const int allowed[] = {1, 2, 3, 4, 6, 8, 12, 16};
    for (int a : allowed)
        if (r <= a) return a;
    return -1;
}
#endif

#if MCD_RGROUP == 0
hipError_t launch_logpdf_g1(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st);
hipError_t launch_logpdf_g2(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st);
hipError_t launch_logpdf_g3(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st);
hipError_t launch_logpdf(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (use_split(M, batch)) return launch_logpdf_split(M, X, ldx, batch, ll, st);
    if (use_wide(M, batch)) return launch_logpdf_wide(M, X, ldx, batch, ll, st);
    if (M.R == 6 || M.R == 8) return launch_logpdf_g1(M, X, ldx, batch, ll, st);
    if (M.R == 12) return launch_logpdf_g2(M, X, ldx, batch, ll, st);
    if (M.R == 16) return launch_logpdf_g3(M, X, ldx, batch, ll, st);
#else
hipError_t MCD_CAT(launch_logpdf_g, MCD_RGROUP)(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st)
{
#endif
#define CALL(R) launch_logpdf_R<R>(M, X, ldx, batch, ll, st)
    MCD_DISPATCH_R(M.R, CALL)
#undef CALL
}

}  // namespace mcd

#ifdef MCD_STAMP
extern "C" int mcd_debug_stamps(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mcd::g_dbg), 64 * sizeof(unsigned long long));
}
#endif
#if defined(MCD_STAMP) && defined(MCD_STAMP_LIGHT) && MCD_RGROUP == 0
// the ring of the last 64 launches: hist [64][8][8], span [64][64][2], *n = launches completed
extern "C" int mcd_debug_hist(unsigned long long* hist, unsigned long long* span, unsigned int* n)
{
    if (hipMemcpyFromSymbol(hist, HIP_SYMBOL(mcd::g_hist), 64 * 8 * 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(span, HIP_SYMBOL(mcd::g_span), 64 * 64 * 2 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(n, HIP_SYMBOL(mcd::g_launch), sizeof(unsigned int));
}
#endif
