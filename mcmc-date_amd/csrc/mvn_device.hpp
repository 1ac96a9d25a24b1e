#pragma once
// mvn_device.hpp -- CDNA4 (gfx950) device code for McmcDate's MVN phylogenetic log-likelihood.
//
// What is computed (reference: app/Probability.hs:166-173, 195-207; app/Tools.hs:36-48;
// lib/Mcmc/Tree/Types.hs:224-233; gradient: app/Probability.hs:361-388 by AD in the reference):
//
//   ll[b] = c + (-1/2) (logdetSigma + || L^-1 (x_b - mu) ||^2),   Sigma = L L^T,  c = -N ln sqrt(2 pi)
//
// Mapping ("column sweep").  A workgroup = CW compute waves + LW loader waves.
//   * A compute wave owns BT chains.  Lane l holds rows l, l+64, ..., l+64(R-1) of each chain's
//     residual in registers (R = ceil(N/64) doubles per chain per lane).
//   * The factor is pre-scaled on the host so that the solve needs no divide on the critical
//     path:  Lt[i][j] = L[i][j] / L[i][i] (i > j),  d~_i = (x_i - mu_i) / L[i][i].
//     Then for j = 0..N-1:  z_j = d~_j (already final);  d~_i -= Lt[i][j] z_j  for i > j.
//   * z_j lives in lane (j mod 64): it is broadcast with two v_readlane_b32 into an SGPR pair
//     and consumed as the scalar operand of v_fma_f64 -- no cross-lane reduction inside the
//     sweep.  The broadcast of column j+1 is issued between the FMAs of column j (measured:
//     27.5 instead of 35 cycles per column at R = 4, tools/microbench/lat2.hip).  The only
//     reduction is the final sum z^2 (DPP + readlane).
//   * Lt is consumed in CHUNKS of CP column pairs through a two-slot LDS ring.  One CU can take in
//     only ~25-45 B/clk of L2-resident data and one wave only ~13-25 B/clk (tools/microbench/
//     stream.hip), and every LDS write costs the writing wave issue time, so dedicated LOADER
//     waves (global -> registers -> ds_write_b128, 16 B per lane, 1 KiB per wave-instruction,
//     three chunks ahead) feed the ring while the compute waves only read it (ds_read_b128,
//     conflict-free) and run the FMA chain.  One s_barrier per chunk for all waves.
//   * Rows on/above the diagonal inside a 64-row block multiply stored zeros (exact for
//     finite data; a non-finite z_j turns the chain's result into NaN, which the sampler
//     rejects exactly like the reference's NaN/Inf -- see DESIGN.md "Non-finite inputs").
//   * The gradient kernels run the mirrored sweep with Ut = scaled L^T from the last column
//     to the first on the same registers:  y = L^-T z = Sigma^-1 (x - mu),  grad_x = -y.
//
// No MFMA on purpose: this is TRSV + DOT (north_star), fp64 FMA on the vector ALU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mvn_kernels.h"
#include "options.h"

namespace mcd {

// native 2 x f64 vector: loads/stores stay first-class values (HIP's double2 class copies through
// memcpy, which pins register arrays to scratch)
typedef double d2 __attribute__((ext_vector_type(2)));


// ---------------------------------------------------------------------------------------
// cross-lane helpers (wave64)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane64(double v, int srclane /* wave-uniform */)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, srclane);
    hi = __builtin_amdgcn_readlane(hi, srclane);
    return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes, result wave-uniform.  Fixed order => bit-reproducible.
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_mov64<0xB1>(v);   // quad_perm [1,0,3,2]  (xor 1)
    v += dpp_mov64<0x4E>(v);   // quad_perm [2,3,0,1]  (xor 2)
    v += dpp_mov64<0x124>(v);  // row_ror:4
    v += dpp_mov64<0x128>(v);  // row_ror:8  -> every lane holds its 16-lane row sum
    return (readlane64(v, 0) + readlane64(v, 16)) + (readlane64(v, 32) + readlane64(v, 48));
}

// All LDS operations of this wave done, then workgroup barrier.  Deliberately NOT __syncthreads():
// that would also drain vmcnt and serialise the global prefetch that is still in flight.
// NOTE: pointers into the LDS ring must never be __restrict__: other waves rewrite the ring between
// barriers, and with a noalias pointer the compiler may keep values loaded from a slot across the
// barrier (seen once: chunk ci+2 computed with chunk ci's factor values when registers allowed it).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#define MCD_SB __builtin_amdgcn_sched_barrier(0)

// Diagnostic builds only (-DMCD_STAMP): wave-level timestamps of workgroup 0 at a few milestones.
// -DMCD_STAMP_LIGHT (with MCD_STAMP; k_logpdf.hip, `make stamp_headline`): the milestones only, kept per launch in a ring of the
// last 64 launches together with entry / exit of every workgroup on workgroup 0's XCD -- the budget of a graph-replayed launch
// (tools/microbench/headline_phases.py); no accumulators inside the sweep, so the loop under test is the shipped one.
#if defined(MCD_STAMP) && defined(MCD_STAMP_LIGHT)
__device__ unsigned long long g_dbg[64];
__device__ unsigned long long g_hist[64][8][8];            // [launch % 64][wave][milestone] of workgroup 0
__device__ unsigned long long g_span[64][64][2];           // [launch % 64][i][entry, exit] of compute wave 0 of workgroup 8 i
__device__ unsigned int g_launch;                          // launches completed (bumped by workgroup 0 at its exit)
#define MCD_T(idx)                                                                               \
    do {                                                                                         \
        unsigned long long t_;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
        if ((threadIdx.x & 63) == 0) {                                                           \
            const unsigned ln_ = __atomic_load_n(&g_launch, __ATOMIC_RELAXED) & 63u;             \
            if (blockIdx.x == 0) g_hist[ln_][threadIdx.x >> 6][(idx)] = t_;                       \
            if ((blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 64 && threadIdx.x == 0 && ((idx) == 0 || (idx) == 4)) \
                g_span[ln_][blockIdx.x >> 3][(idx) == 0 ? 0 : 1] = t_;                           \
            if (blockIdx.x == 0 && threadIdx.x == 0 && (idx) == 4) __atomic_fetch_add(&g_launch, 1u, __ATOMIC_RELAXED); \
        }                                                                                        \
    } while (0)
#define MCD_ACC_DECL
#define MCD_ACC(idx) do { } while (0)
#define MCD_ACC_PARAMS
#define MCD_ACC_ARGS
#define MCD_ACC_FLUSH(base) do { } while (0)
#elif defined(MCD_STAMP)
__device__ unsigned long long g_dbg[64];
#define MCD_T(idx)                                                                               \
    do {                                                                                         \
        unsigned long long t_;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_dbg[(threadIdx.x >> 6) * 8 + (idx)] = t_; \
    } while (0)
// accumulate the cycles since the previous MCD_ACC of this wave into bucket idx (idx < 0: restart)
__device__ __forceinline__ void mcd_acc(int idx, unsigned long long& tprev, unsigned long long* acc)
{
    unsigned long long t_;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    if (idx >= 0) acc[idx] += t_ - tprev;
    tprev = t_;
}
#define MCD_ACC_DECL unsigned long long tprev_ = 0, acc_[4] = {0, 0, 0, 0};
#define MCD_ACC(idx) mcd_acc(idx, tprev_, acc_)
#define MCD_ACC_PARAMS , unsigned long long& tprev_, unsigned long long* acc_
#define MCD_ACC_ARGS , tprev_, acc_
#define MCD_ACC_FLUSH(base)                                                                      \
    do {                                                                                         \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0)                                          \
            for (int i_ = 0; i_ < 3; ++i_) g_dbg[(threadIdx.x >> 6) * 8 + (base) + i_] = acc_[i_]; \
    } while (0)
#else
#define MCD_T(idx) do { } while (0)
#define MCD_ACC_DECL
#define MCD_ACC(idx) do { } while (0)
#define MCD_ACC_PARAMS
#define MCD_ACC_ARGS
#define MCD_ACC_FLUSH(base) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------
// Factor stream.  Element (row = 64k + lane, column j) of a scaled triangular factor padded to
// NP = 64 R lives at   (((j >> 1) * R + k) * 64 + lane) * 2 + (j & 1)    ("pair-interleaved
// column layout": one 16-byte load per lane fetches two adjacent columns, a wave-instruction
// reads one contiguous 1-KiB UNIT = (column pair, row block)).
// ---------------------------------------------------------------------------------------
template <int R>
struct Cfg {
    // 1-KiB units per LDS ring slot (two slots): 32 KiB slots up to N = 512; above that a chunk of 32 units is
    // only 4 columns and the per-chunk barrier + loader bookkeeping dominate (measured at N = 1024: the compute
    // waves spent half of the sweep waiting at barriers), so 64-unit slots (128 KiB of LDS, one workgroup per CU).
    static constexpr int SU = (R >= 12) ? 64 : 32;
    // loader waves per workgroup
    static constexpr int LW = (R >= 12) ? 4 : 2;
    // column pairs per chunk: CP * R <= SU
    static constexpr int CP = (R == 1) ? 32 : (R == 2) ? 16 : (R <= 4) ? 8 : (R <= 8) ? 4 : (SU / 16);
    static constexpr int CPB = 32 / CP;                     // chunks per 64-column block
    static constexpr int NCHUNK = R * CPB;
    static constexpr int CCOLS = 2 * CP;                    // columns per chunk
};

constexpr int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// Staging registers of one loader wave: its share (units u = i * LW + lw) of the chunks in flight.
// Set s holds a chunk of parity s.
template <int R, int LW>
struct Stage {
    static constexpr int MAXU = (Cfg<R>::CP * R + LW - 1) / LW;
    d2 v[2][MAXU];
};

// chunk of a sweep: column pairs [p0, p0 + CP), row blocks [KLO, KHI).  Unit u of the chunk goes to
// loader u % LW.  When the unit count is not a multiple of LW the surplus slots of the schedule
// reload the last unit (harmless) and skip the LDS store.
template <int R, int LW, int KLO, int KHI, int SET>
__device__ __forceinline__ void stage_load(Stage<R, LW>& st, const double* __restrict__ F, int p0, int lw, int lane)
{
    constexpr int NK = KHI - KLO;
    constexpr int NU = Cfg<R>::CP * NK;
#pragma unroll
    for (int i = 0; i < (NU + LW - 1) / LW; ++i) {
        int u = i * LW + lw;                              // wave-uniform
        if constexpr (NU % LW != 0) u = u < NU ? u : NU - 1;
        const int p = p0 + u / NK, k = KLO + u % NK;
        st.v[SET][i] = reinterpret_cast<const d2*>(F)[((size_t)p * R + k) * 64 + lane];
    }
}
// (Measured and dropped in round 4: the lanes of a DIAGONAL row block's unit that hold stored zeros -- rows up to the unit's first column,
// about a fifth of the whole stream at R = 4 -- not loading but staging the zero themselves.  Fewer bytes, but the exec-masked loads and the
// selects in the loaders' issue stream cost far more than the bytes saved: 7.6 -> 13.5 us per launch at N = 256 x 512 chains, same bits.)

template <int R, int LW, int KLO, int KHI, int SET>
__device__ __forceinline__ void stage_store(const Stage<R, LW>& st, d2* slot, int lw, int lane)
{
    constexpr int NU = Cfg<R>::CP * (KHI - KLO);
#pragma unroll
    for (int i = 0; i < (NU + LW - 1) / LW; ++i) {
        const int u = i * LW + lw;
        if constexpr (NU % LW != 0) {
            if (u < NU) slot[u * 64 + lane] = st.v[SET][i];
        } else {
            slot[u * 64 + lane] = st.v[SET][i];
        }
    }
}

// LDS -> VGPR prefetch distance (column pairs).  One wave per SIMD has nothing else to overlap the
// ~100-cycle LDS latency with, so the factor values of pair p + LDS_PD are requested before pair p
// is applied.
// (At R = 16 a prefetch buffer is 16 row blocks deep: one pair ahead there -- with two the gradient kernels, which hold both sweeps'
// buffers, spill 200 to 1 200 registers.  The sweeps are the fallback at that size: the row split and the multiply form serve it.)
template <int R> struct LdsPd { static constexpr int PD = (R >= 16) ? 1 : 2; };

// =======================================================================================
// forward sweep
// =======================================================================================
// Apply one chunk (CP column pairs starting at column offset jj0 of block JB) from an LDS slot.
template <int R, int BT, int JB, int JJ0>
__device__ __forceinline__ void fwd_apply(double (&d)[R][BT], const d2* slot, int lane)
{
    constexpr int jj0 = JJ0;
    constexpr int CP = Cfg<R>::CP;
    constexpr int NK = R - JB;
    constexpr int LDS_PD = LdsPd<R>::PD;
    constexpr int NBUF = LDS_PD + 1;
    d2 l[NBUF][NK];
#pragma unroll
    for (int p = 0; p < LDS_PD && p < CP; ++p)
#pragma unroll
        for (int k = 0; k < NK; ++k) l[p % NBUF][k] = slot[(p * NK + k) * 64 + lane];
    double z[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) z[b] = readlane64(d[JB][b], jj0);
#pragma unroll
    for (int p = 0; p < CP; ++p) {
        if (p + LDS_PD < CP) {
#pragma unroll
            for (int k = 0; k < NK; ++k) l[(p + LDS_PD) % NBUF][k] = slot[((p + LDS_PD) * NK + k) * 64 + lane];
        }
        MCD_SB;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int jj = jj0 + 2 * p + h;
            const bool last = (p == CP - 1) && (h == 1);
            // row block JB first: it holds row j+1, whose value is the next column's z
#pragma unroll
            for (int b = 0; b < BT; ++b) d[JB][b] = fma(-(h ? l[p % NBUF][0].y : l[p % NBUF][0].x), z[b], d[JB][b]);
            MCD_SB;
            if constexpr (NK > 1) {
#pragma unroll
                for (int b = 0; b < BT; ++b)
                    d[JB + 1][b] = fma(-(h ? l[p % NBUF][1].y : l[p % NBUF][1].x), z[b], d[JB + 1][b]);
                MCD_SB;
            }
            double zn[BT];
            if (!last) {
#pragma unroll
                for (int b = 0; b < BT; ++b) zn[b] = readlane64(d[JB][b], jj + 1);   // broadcast for column j+1
            }
            MCD_SB;
#pragma unroll
            for (int k = 2; k < NK; ++k) {
#pragma unroll
                for (int b = 0; b < BT; ++b)
                    d[JB + k][b] = fma(-(h ? l[p % NBUF][k].y : l[p % NBUF][k].x), z[b], d[JB + k][b]);
            }
            MCD_SB;
            if (!last) {
#pragma unroll
                for (int b = 0; b < BT; ++b) z[b] = zn[b];
            }
        }
    }
}

// ---- compute role -------------------------------------------------------------------------
// The chunk index is a compile-time constant so that every v_readlane has an immediate lane
// select (a lane select in a freshly written SGPR costs ~9 cycles per column, lat2.hip).
template <int R, int BT, int JB, int LC>
__device__ __forceinline__ bool fwd_compute_chunks(double (&d)[R][BT], const d2* ring, int lane, int ncols MCD_ACC_PARAMS)
{
    using C = Cfg<R>;
    if constexpr (LC < C::CPB) {
        constexpr int CI = JB * C::CPB + LC;
        if (CI * C::CCOLS >= ncols) return false;        // workgroup-uniform
        MCD_ACC(-1);
        fwd_apply<R, BT, JB, LC * C::CCOLS>(d, ring + (CI & 1) * C::SU * 64, lane);
        MCD_ACC(0);
        lds_barrier();
        MCD_ACC(1);
        return fwd_compute_chunks<R, BT, JB, LC + 1>(d, ring, lane, ncols MCD_ACC_ARGS);
    } else {
        return true;
    }
}

template <int R, int BT, int JB>
__device__ __forceinline__ void fwd_compute(double (&d)[R][BT], const d2* ring, int lane, int ncols MCD_ACC_PARAMS)
{
    if constexpr (JB < R) {
        if (!fwd_compute_chunks<R, BT, JB, 0>(d, ring, lane, ncols MCD_ACC_ARGS)) return;
        fwd_compute<R, BT, JB + 1>(d, ring, lane, ncols MCD_ACC_ARGS);
    }
}

// ---- loader role --------------------------------------------------------------------------
// Chunk ci = JB * CPB + lc.  On entry staging set SLOT^1 holds the loader's share of chunk ci+1
// (block JB1) and set SLOT that of chunk ci+2 (both in flight); chunk ci+3 (block JB3) is requested
// here.  The LDS write of chunk ci+1 goes to slot SLOT^1, which every compute wave has finished
// reading (barrier of chunk ci-1).
template <int R, int LW, int JB, int JB1, int JB3, int SLOT>
__device__ __forceinline__ bool fwd_loader_chunk(const double* __restrict__ Ft, d2* ring, Stage<R, LW>& st,
                                                 int lw, int lane, int ncols, int lc MCD_ACC_PARAMS)
{
    using C = Cfg<R>;
    const int ci = JB * C::CPB + lc;
    if (ci * C::CCOLS >= ncols) return false;
    MCD_ACC(-1);
    if constexpr (JB1 < R) stage_store<R, LW, JB1, R, SLOT ^ 1>(st, ring + (SLOT ^ 1) * C::SU * 64, lw, lane);
    MCD_ACC(0);
    if constexpr (JB3 < R) stage_load<R, LW, JB3, R, SLOT ^ 1>(st, Ft, (ci + 3) * C::CP, lw, lane);
    MCD_ACC(2);
    lds_barrier();
    MCD_ACC(1);
    return true;
}

template <int R, int LW, int JB, int LC>
__device__ __forceinline__ bool fwd_loader_tail(const double* __restrict__ Ft, d2* ring, Stage<R, LW>& st,
                                                int lw, int lane, int ncols MCD_ACC_PARAMS)
{
    constexpr int CPB = Cfg<R>::CPB;
    if constexpr (LC < CPB) {
        constexpr int JB1 = JB + (LC + 1) / CPB;
        constexpr int JB3 = JB + (LC + 3) / CPB;
        constexpr int SLOT = (JB * CPB + LC) & 1;
        if (!fwd_loader_chunk<R, LW, JB, JB1, JB3, SLOT>(Ft, ring, st, lw, lane, ncols, LC MCD_ACC_ARGS)) return false;
        return fwd_loader_tail<R, LW, JB, LC + 1>(Ft, ring, st, lw, lane, ncols MCD_ACC_ARGS);
    } else {
        return true;
    }
}

template <int R, int LW, int JB>
__device__ __forceinline__ void fwd_loader(const double* __restrict__ Ft, d2* ring, Stage<R, LW>& st, int lw,
                                           int lane, int ncols MCD_ACC_PARAMS)
{
    constexpr int CPB = Cfg<R>::CPB;
    if constexpr (JB < R) {
        constexpr int T0 = CPB >= 4 ? CPB - 4 : 0;          // first chunk of the compile-time tail
        for (int lc = 0; lc + 4 < CPB; lc += 2) {           // interior: the next three chunks are in this block
            if (!fwd_loader_chunk<R, LW, JB, JB, JB, 0>(Ft, ring, st, lw, lane, ncols, lc MCD_ACC_ARGS)) return;
            if (!fwd_loader_chunk<R, LW, JB, JB, JB, 1>(Ft, ring, st, lw, lane, ncols, lc + 1 MCD_ACC_ARGS)) return;
        }
        if (!fwd_loader_tail<R, LW, JB, T0>(Ft, ring, st, lw, lane, ncols MCD_ACC_ARGS)) return;
        fwd_loader<R, LW, JB + 1>(Ft, ring, st, lw, lane, ncols MCD_ACC_ARGS);
    }
}

// Start of the stream.  Only chunk 0 is requested before the first barrier: the compute waves' state loads
// share the CU's memory pipeline with the loaders, and a deeper initial burst (chunks 0-2 = 96 KiB)
// held the first barrier back by ~1.5k cycles.  Chunks 1 and 2 are requested right after the barrier
// (fwd_loader_start), while the compute waves apply chunk 0.
template <int R, int LW>
__device__ __forceinline__ void fwd_loader_prologue(const double* __restrict__ Ft, d2* ring, Stage<R, LW>& st, int lw,
                                                    int lane)
{
    stage_load<R, LW, 0, R, 0>(st, Ft, 0, lw, lane);
    stage_store<R, LW, 0, R, 0>(st, ring, lw, lane);
}

template <int R, int LW>
__device__ __forceinline__ void fwd_loader_start(const double* __restrict__ Ft, Stage<R, LW>& st, int lw, int lane)
{
    using C = Cfg<R>;
    if constexpr (C::NCHUNK > 1) stage_load<R, LW, 1 / C::CPB, R, 1>(st, Ft, C::CP, lw, lane);
    if constexpr (C::NCHUNK > 2) stage_load<R, LW, 2 / C::CPB, R, 0>(st, Ft, 2 * C::CP, lw, lane);
}

// =======================================================================================
// backward sweep.  Ut holds the scaled transpose: element (row r, column i) = L[i][r] / L[r][r] for
// r < i, zero otherwise; column block IB touches row blocks k <= IB.  Chunks are visited from the
// one that holds column ncols-1 down to chunk 0, columns inside a chunk from high to low.
// =======================================================================================
template <int R, int BT, int IB, int II0>
__device__ __forceinline__ void bwd_apply(double (&d)[R][BT], const d2* slot, int lane)
{
    constexpr int ii0 = II0;
    constexpr int CP = Cfg<R>::CP;
    constexpr int NK = IB + 1;
    constexpr int LDS_PD = LdsPd<R>::PD;
    constexpr int NBUF = LDS_PD + 1;
    d2 u[NBUF][NK];
    // pairs are visited CP-1, CP-2, ..., 0; q counts visited pairs
#pragma unroll
    for (int q = 0; q < LDS_PD && q < CP; ++q)
#pragma unroll
        for (int k = 0; k < NK; ++k) u[q % NBUF][k] = slot[((CP - 1 - q) * NK + k) * 64 + lane];
    double y[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) y[b] = readlane64(d[IB][b], ii0 + 2 * CP - 1);
#pragma unroll
    for (int q = 0; q < CP; ++q) {
        const int p = CP - 1 - q;
        if (q + LDS_PD < CP) {
#pragma unroll
            for (int k = 0; k < NK; ++k) u[(q + LDS_PD) % NBUF][k] = slot[((p - LDS_PD) * NK + k) * 64 + lane];
        }
        MCD_SB;
#pragma unroll
        for (int h = 1; h >= 0; --h) {
            const int ii = ii0 + 2 * p + h;
            const bool last = (p == 0) && (h == 0);
            // row block IB first: it holds row i-1, whose value is the next column's y
#pragma unroll
            for (int b = 0; b < BT; ++b) d[IB][b] = fma(-(h ? u[q % NBUF][IB].y : u[q % NBUF][IB].x), y[b], d[IB][b]);
            MCD_SB;
            if constexpr (NK > 1) {
#pragma unroll
                for (int b = 0; b < BT; ++b)
                    d[IB - 1][b] = fma(-(h ? u[q % NBUF][IB - 1].y : u[q % NBUF][IB - 1].x), y[b], d[IB - 1][b]);
                MCD_SB;
            }
            double yn[BT];
            if (!last) {
#pragma unroll
                for (int b = 0; b < BT; ++b) yn[b] = readlane64(d[IB][b], ii - 1);
            }
            MCD_SB;
#pragma unroll
            for (int k = IB - 2; k >= 0; --k) {
#pragma unroll
                for (int b = 0; b < BT; ++b) d[k][b] = fma(-(h ? u[q % NBUF][k].y : u[q % NBUF][k].x), y[b], d[k][b]);
            }
            MCD_SB;
            if (!last) {
#pragma unroll
                for (int b = 0; b < BT; ++b) y[b] = yn[b];
            }
        }
    }
}

// ---- compute role -------------------------------------------------------------------------
// `started` is false until the top chunk (the first one with columns < ncols) is reached; the
// loaders publish it with one extra barrier that the compute waves match here.
template <int R, int BT, int IB, int LC>
__device__ __forceinline__ void bwd_compute_chunks(double (&d)[R][BT], const d2* ring, int lane, int ncols,
                                                   bool& started)
{
    using C = Cfg<R>;
    if constexpr (LC >= 0) {
        constexpr int CI = IB * C::CPB + LC;
        if (CI * C::CCOLS < ncols) {                       // workgroup-uniform
            if (!started) {
                lds_barrier();                             // top chunk published by the loaders
                started = true;
            }
            bwd_apply<R, BT, IB, LC * C::CCOLS>(d, ring + (CI & 1) * C::SU * 64, lane);
            lds_barrier();
        }
        bwd_compute_chunks<R, BT, IB, LC - 1>(d, ring, lane, ncols, started);
    }
}

template <int R, int BT, int IB>
__device__ __forceinline__ void bwd_compute(double (&d)[R][BT], const d2* ring, int lane, int ncols,
                                            bool& started)
{
    if constexpr (IB >= 0) {
        bwd_compute_chunks<R, BT, IB, Cfg<R>::CPB - 1>(d, ring, lane, ncols, started);
        bwd_compute<R, BT, IB - 1>(d, ring, lane, ncols, started);
    }
}

// ---- loader role --------------------------------------------------------------------------
template <int R, int LW, int IB, int IB1, int IB2, int IB3, int SLOT>
__device__ __forceinline__ void bwd_loader_chunk(const double* __restrict__ Ut, d2* ring, Stage<R, LW>& st,
                                                 int lw, int lane, int ncols, int lc, bool& started)
{
    using C = Cfg<R>;
    const int ci = IB * C::CPB + lc;
    if (ci * C::CCOLS >= ncols) return;                  // above the swept columns (workgroup-uniform)
    if (!started) {                                        // top chunk: fill the pipeline
        stage_load<R, LW, 0, IB + 1, SLOT>(st, Ut, ci * C::CP, lw, lane);
        if constexpr (IB1 >= 0) stage_load<R, LW, 0, IB1 + 1, SLOT ^ 1>(st, Ut, (ci - 1) * C::CP, lw, lane);
        stage_store<R, LW, 0, IB + 1, SLOT>(st, ring + SLOT * C::SU * 64, lw, lane);
        if constexpr (IB2 >= 0) stage_load<R, LW, 0, IB2 + 1, SLOT>(st, Ut, (ci - 2) * C::CP, lw, lane);
        lds_barrier();
        started = true;
    }
    if constexpr (IB1 >= 0) stage_store<R, LW, 0, IB1 + 1, SLOT ^ 1>(st, ring + (SLOT ^ 1) * C::SU * 64, lw, lane);
    if constexpr (IB3 >= 0) stage_load<R, LW, 0, IB3 + 1, SLOT ^ 1>(st, Ut, (ci - 3) * C::CP, lw, lane);
    lds_barrier();
}

template <int R, int LW, int IB, int LC>
__device__ __forceinline__ void bwd_loader_tail(const double* __restrict__ Ut, d2* ring, Stage<R, LW>& st,
                                                int lw, int lane, int ncols, bool& started)
{
    constexpr int CPB = Cfg<R>::CPB;
    if constexpr (LC >= 0) {
        constexpr int IB1 = IB + floordiv(LC - 1, CPB);
        constexpr int IB2 = IB + floordiv(LC - 2, CPB);
        constexpr int IB3 = IB + floordiv(LC - 3, CPB);
        constexpr int SLOT = (IB * CPB + LC) & 1;
        bwd_loader_chunk<R, LW, IB, IB1, IB2, IB3, SLOT>(Ut, ring, st, lw, lane, ncols, LC, started);
        bwd_loader_tail<R, LW, IB, LC - 1>(Ut, ring, st, lw, lane, ncols, started);
    }
}

template <int R, int LW, int IB>
__device__ __forceinline__ void bwd_loader(const double* __restrict__ Ut, d2* ring, Stage<R, LW>& st, int lw,
                                           int lane, int ncols, bool& started)
{
    constexpr int CPB = Cfg<R>::CPB;
    if constexpr (IB >= 0) {
        for (int lc = CPB - 1; lc >= 5; lc -= 2) {         // interior: the next three chunks are in this block
            bwd_loader_chunk<R, LW, IB, IB, IB, IB, 1>(Ut, ring, st, lw, lane, ncols, lc, started);
            bwd_loader_chunk<R, LW, IB, IB, IB, IB, 0>(Ut, ring, st, lw, lane, ncols, lc - 1, started);
        }
        bwd_loader_tail<R, LW, IB, (CPB >= 4 ? 3 : CPB - 1)>(Ut, ring, st, lw, lane, ncols, started);
        bwd_loader<R, LW, IB - 1>(Ut, ring, st, lw, lane, ncols, started);
    }
}

// ---------------------------------------------------------------------------------------
// state -> residual prologues (compute waves)
// ---------------------------------------------------------------------------------------
template <int R, int BT>
__device__ __forceinline__ void load_rawx(double (&d)[R][BT], const MvnDev& M, const double* __restrict__ X, int64_t ldx,
                                          int64_t b0, int64_t batch, int lane)
{
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int row = 64 * k + lane;
        const double m = M.mu[row];        // padded: 0
        const double iv = M.invdiag[row];  // padded: 1
#pragma unroll
        for (int c = 0; c < BT; ++c) {
            const int64_t b = b0 + c;
            double xv = m;
            if (row < M.n && b < batch) xv = X[b * ldx + row];
            d[k][c] = (xv - m) * iv;       // dxs = xs - mu  (app/Probability.hs:171), then row scaling
        }
    }
}

// distances from the tree state -- app/Probability.hs:201-207 with app/Tools.hs:36-48 and
// lib/Mcmc/Tree/Types.hs:224-233 folded into index tables (slot -> node, node -> parent).
template <int R, int BT>
__device__ __forceinline__ void load_tree(double (&d)[R][BT], double (&dist)[R][BT], const MvnDev& M, const TreeDev& T,
                                          const double* __restrict__ H, const double* __restrict__ Rt, int64_t lds,
                                          const double* __restrict__ tH, const double* __restrict__ rMu, int64_t b0,
                                          int64_t batch, int lane)
{
    double s[BT];
#pragma unroll
    for (int c = 0; c < BT; ++c) {
        const int64_t b = (b0 + c < batch) ? b0 + c : batch - 1;
        s[c] = tH[b] * rMu[b];             // :205-207  (tH * rMu)
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int row = 64 * k + lane;
        const double m = M.mu[row];
        const double iv = M.invdiag[row];
        const int a = T.slot_node[row];    // -1 for padded rows
        const int pa = T.slot_parent[row]; // 0 for padded rows
#pragma unroll
        for (int c = 0; c < BT; ++c) {
            const int64_t b = (b0 + c < batch) ? b0 + c : batch - 1;
            const double* h = H + b * lds;
            const double* r = Rt + b * lds;
            double v = 0.0;
            if (a >= 0) {
                v = (h[pa] - h[a]) * r[a];                           // zipWith (*) times rates
                if (row == 0) v = v + (h[0] - h[T.root_right]) * r[T.root_right];  // sumFirstTwo
                v = v * s[c];                                         // map (* (tH * rMu))
            }
            dist[k][c] = v;
            d[k][c] = (v - m) * iv;
        }
    }
}

// The same with the chain's height and rate rows passed through LDS (`stage`: 2 (64 R + 64) doubles private to the wave): the
// rows are read contiguously, side by side with the slot tables -- one round trip -- and the gather by node id happens in LDS,
// where load_tree gathers from global memory once the tables have arrived: two dependent round trips and a line per lane.
// Same arithmetic, same bits.  One chain per wave (BT = 1).
template <int R>
__device__ __forceinline__ void load_tree_staged(double (&d)[R][1], double (&dist)[R][1], const MvnDev& M, const TreeDev& T,
                                                 const double* __restrict__ H, const double* __restrict__ Rt, int64_t lds,
                                                 const double* __restrict__ tH, const double* __restrict__ rMu, int64_t b0,
                                                 int64_t batch, int lane, double* stage)
{
    constexpr int NP = 64 * R + 64;
    const int64_t b = (b0 < batch) ? b0 : batch - 1;
    const double* __restrict__ h = H + b * lds;
    const double* __restrict__ r = Rt + b * lds;
    const int nn = T.n_nodes;                              // <= 64 R + 2
    double hv[R + 1], rv[R + 1], m[R], iv[R];
    int a[R], pa[R];
#pragma unroll
    for (int i = 0; i <= R; ++i) {
        const int v = 64 * i + lane;
        const int vc = v < nn ? v : nn - 1;
        hv[i] = h[vc];
        rv[i] = r[vc];
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int row = 64 * k + lane;
        m[k] = M.mu[row];
        iv[k] = M.invdiag[row];
        a[k] = T.slot_node[row];           // -1 for padded rows
        pa[k] = T.slot_parent[row];        // 0 for padded rows
    }
    const double s = tH[b] * rMu[b];       // :205-207  (tH * rMu)
    const int rr = T.root_right;
#pragma unroll
    for (int i = 0; i <= R; ++i) {
        stage[64 * i + lane] = hv[i];
        stage[NP + 64 * i + lane] = rv[i];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): the rows are in LDS (one wave: LDS operations complete in order)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int row = 64 * k + lane;
        double v = 0.0;
        if (a[k] >= 0) {
            v = (stage[pa[k]] - stage[a[k]]) * stage[NP + a[k]];                           // zipWith (*) times rates
            if (row == 0) v = v + (stage[0] - stage[rr]) * stage[NP + rr];                 // sumFirstTwo
            v = v * s;                                                                      // map (* (tH * rMu))
        }
        dist[k][0] = v;
        d[k][0] = (v - m[k]) * iv[k];
    }
}

template <int R, int BT>
__device__ __forceinline__ void finish_ll(const double (&d)[R][BT], const MvnDev& M, int64_t b0, int64_t batch,
                                          double* __restrict__ ll, int lane)
{
#pragma unroll
    for (int c = 0; c < BT; ++c) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) s = fma(d[k][c], d[k][c], s);
        const double q = wave_sum(s);
        if (lane == 0 && b0 + c < batch) ll[b0 + c] = M.c + (-0.5) * (M.logdet + q);  // app/Probability.hs:169
    }
}

// ---------------------------------------------------------------------------------------
// Kernel skeleton.  Waves 0 .. CW-1 compute, waves CW .. CW+LW-1 load.  Every wave of a
// workgroup takes part in every barrier, also compute waves whose chains lie beyond the batch
// (they work on clamped inputs and store nothing): no early exits before the last barrier.
// ---------------------------------------------------------------------------------------
// (MCD_BID: the workgroup's index among the likelihood workgroups; a kernel that puts workgroups of another role in front of
// them -- k_tree_logpdf.hip -- defines it before including this header)
#ifndef MCD_BID
#define MCD_BID blockIdx.x
#endif
#define MCD_KERNEL_HEAD                                                         \
    __shared__ d2 ring[2 * Cfg<R>::SU * 64];                                    \
    const int lane = threadIdx.x & 63;                                          \
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          \
    const int64_t b0 = ((int64_t)MCD_BID * CW + wave) * BT;                      \
    const int ncols = M.ncols;   /* swept columns: N rounded up to whole chunks (host: sweep_chunk_columns) */

// launch geometry by batch size (host side)
struct Geometry {
    int cw, lw, bt;
};
static inline Geometry pick_geometry(int64_t batch)
{
    // <= 512 chains (a sampler's usual batch): 2 compute waves + 2 loaders per workgroup, so that every
    // chain gets a SIMD to itself and all 256 CUs take part in pulling the factor out of L2.
    // Up to 4096 chains: 4 compute waves per workgroup keep the grid within one wave of workgroups
    // per CU for longer (measured at N = 256, B = 1024: 9.5 us against 14.4 us).
    // More: 4 compute waves x 2 chains share each pass over the factor.
    const int force = opt_get(OPT_GEOM);                 // tuning (mcd_set_option "MCD_GEOM"): 21 | 41 | 42 = compute waves, chains per wave
    if (force == 21) return {2, 2, 1};
    if (force == 41 || force == 42) return {4, 2, force == 42 ? 2 : 1};
    if (batch <= 512) return {2, 2, 1};
    if (batch <= 4096) return {4, 2, 1};
    return {4, 2, 2};
}

}  // namespace mcd
