// mvn_kernels.hip -- CDNA4 (gfx950) kernels for McmcDate's MVN phylogenetic log-likelihood.
//
// What is computed (reference: app/Probability.hs:166-173, 195-207; app/Tools.hs:36-48;
// lib/Mcmc/Tree/Types.hs:224-233; gradient: app/Probability.hs:361-388 by AD in the reference):
//
//   ll[b] = c + (-1/2) (logdetSigma + || L^-1 (x_b - mu) ||^2),   Sigma = L L^T,  c = -N ln sqrt(2 pi)
//
// Mapping ("column sweep", one wave = BT chains, lanes = rows):
//   * A wave owns BT chains.  Lane l holds rows l, l+64, ..., l+64(R-1) of each chain's
//     residual in registers (R = ceil(N/64) doubles per chain per lane).
//   * The factor is pre-scaled on the host so that the solve needs no divide on the
//     critical path:  Lt[i][j] = L[i][j] / L[i][i] (i > j),  d~_i = (x_i - mu_i) / L[i][i].
//     Then for j = 0..N-1:  z_j = d~_j (already final);  d~_i -= Lt[i][j] z_j  for i > j.
//   * z_j lives in lane (j mod 64): it is broadcast with two v_readlane_b32 into an SGPR
//     pair and consumed as the scalar operand of v_fma_f64 -- no LDS, no cross-lane
//     reduction inside the sweep.  The only reduction is the final sum z^2 (DPP + readlane).
//   * Lt is streamed from L2 in the exact order of consumption ("pair-interleaved column
//     layout", see pack_index) with 16-byte per-lane loads, double-buffered in registers.
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

namespace mcd {

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

// ---------------------------------------------------------------------------------------
// Packed factor access.  Element (row = 64k + lane, column j) of a scaled triangular factor
// padded to NP = 64 R lives at   (((j >> 1) * R + k) * 64 + lane) * 2 + (j & 1)
// so that one 16-byte load per lane fetches two adjacent columns and a wave-instruction
// reads 1 KiB contiguously.
// ---------------------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ double2 load_pair(const double* __restrict__ F, int pair, int k, int lane)
{
    const double2* p = reinterpret_cast<const double2*>(F) + ((size_t)pair * R + k) * 64 + lane;
    return *p;
}

// Streaming schedule.  The factor is consumed in CHUNKS of PAIRS column pairs; a ring of NB
// register chunks keeps NB-1 chunks of loads in flight ahead of the FMAs (one wave has nobody
// else to hide its L2 latency behind at the batch sizes a sampler uses: measured ~800 cycles per
// 16-KiB burst, MI355X).  NB * 2 * PAIRS columns are consumed per loop trip and must divide 64.
template <int R>
struct Cfg {
    static constexpr int PAIRS = (R <= 1) ? 4 : (R <= 4) ? 2 : 1;
    static constexpr int CH = 2 * PAIRS;    // columns per chunk
    static constexpr int NB = (R <= 8) ? 8 : 4;   // ring depth: <= 64 loads (256 VGPRs) in flight
    static constexpr int NP = 64 * R;
};

// One chunk = PAIRS column pairs x row blocks [KLO, KHI).
template <int R, int KLO, int KHI>
struct Chunk {
    double2 v[Cfg<R>::PAIRS][KHI - KLO];
};
template <int R, int KLO, int KHI, int NB>
struct Ring {
    Chunk<R, KLO, KHI> b[NB];
};

template <int R, int KLO, int KHI>
__device__ __forceinline__ void chunk_load(Chunk<R, KLO, KHI>& c, const double* __restrict__ F, int pair_first, int step,
                                           int lane)
{
    // pairs pair_first, pair_first + step, ... (step = +1 forward, -1 backward), clamped in bounds
    constexpr int PAIRS = Cfg<R>::PAIRS;
    constexpr int MAXPAIR = 32 * R - 1;
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
        int pr = pair_first + step * p;
        pr = pr > MAXPAIR ? MAXPAIR : (pr < 0 ? 0 : pr);   // redundant in-bounds reload past either end
#pragma unroll
        for (int k = KLO; k < KHI; ++k) c.v[p][k - KLO] = load_pair<R>(F, pr, k, lane);
    }
}

// ------------------------------- forward sweep ----------------------------------------
template <int R, int BT, int JB>
__device__ __forceinline__ void fwd_apply(double (&d)[R][BT], const Chunk<R, JB, R>& c, int jj0 /* column offset inside block JB */)
{
    constexpr int PAIRS = Cfg<R>::PAIRS;
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int jj = jj0 + 2 * p + h;
            double z[BT];
#pragma unroll
            for (int b = 0; b < BT; ++b) z[b] = readlane64(d[JB][b], jj);
#pragma unroll
            for (int k = JB; k < R; ++k) {
                const double l = h ? c.v[p][k - JB].y : c.v[p][k - JB].x;
#pragma unroll
                for (int b = 0; b < BT; ++b) d[k][b] = fma(-l, z[b], d[k][b]);
            }
        }
    }
}

// Column block JB of the forward sweep.  `ring` arrives holding the block's first NB chunks
// (loaded by the kernel prologue for JB = 0, by the previous block's prefetches otherwise).
template <int R, int BT, int NB, int JB>
__device__ __forceinline__ void fwd_block(double (&d)[R][BT], const double* __restrict__ Ft, int lane, int ncols,
                                          Ring<R, JB, R, NB>& ring)
{
    constexpr int PAIRS = Cfg<R>::PAIRS;
    constexpr int CH = Cfg<R>::CH;
    if (64 * JB < ncols) {                  // wave-uniform
        int lim = ncols - 64 * JB;          // multiple of NB * CH
        lim = lim > 64 ? 64 : lim;
        for (int jj0 = 0; jj0 < lim; jj0 += NB * CH) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                fwd_apply<R, BT, JB>(d, ring.b[i], jj0 + i * CH);
                __builtin_amdgcn_sched_barrier(0);   // keep the refill right behind its consumer
                chunk_load<R, JB, R>(ring.b[i], Ft, 32 * JB + ((jj0 + (i + NB) * CH) >> 1), 1, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (JB + 1 < R) {
            Ring<R, JB + 1, R, NB> next;
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int p = 0; p < PAIRS; ++p)
#pragma unroll
                    for (int k = JB + 1; k < R; ++k) next.b[i].v[p][k - JB - 1] = ring.b[i].v[p][k - JB];
            fwd_block<R, BT, NB, JB + 1>(d, Ft, lane, ncols, next);
        }
    }
}

template <int R, int BT, int NB>
__device__ __forceinline__ void fwd_prologue(Ring<R, 0, R, NB>& ring, const double* __restrict__ Ft, int lane)
{
#pragma unroll
    for (int i = 0; i < NB; ++i) chunk_load<R, 0, R>(ring.b[i], Ft, i * Cfg<R>::PAIRS, 1, lane);
}

// ------------------------------- backward sweep ---------------------------------------
// Ut holds the scaled transpose: element (row r, column i) = L[i][r] / L[r][r] for r < i, zero
// otherwise; column block IB touches row blocks k <= IB.  Columns are visited from high to low.
template <int R, int BT, int IB>
__device__ __forceinline__ void bwd_apply(double (&d)[R][BT], const Chunk<R, 0, IB + 1>& c, int ii_hi /* highest (odd) column offset of the chunk */)
{
    constexpr int PAIRS = Cfg<R>::PAIRS;
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
#pragma unroll
        for (int h = 1; h >= 0; --h) {
            const int ii = ii_hi - 2 * p - (1 - h);
            double y[BT];
#pragma unroll
            for (int b = 0; b < BT; ++b) y[b] = readlane64(d[IB][b], ii);
#pragma unroll
            for (int k = 0; k <= IB; ++k) {
                const double u = h ? c.v[p][k].y : c.v[p][k].x;
#pragma unroll
                for (int b = 0; b < BT; ++b) d[k][b] = fma(-u, y[b], d[k][b]);
            }
        }
    }
}

// Column block IB of the backward sweep (blocks are visited from the top one down).  The top
// active block fills the ring itself; lower blocks receive it prefetched.
template <int R, int BT, int NB, int IB>
__device__ __forceinline__ void bwd_block(double (&d)[R][BT], const double* __restrict__ Ut, int lane, int ncols,
                                          Ring<R, 0, IB + 1, NB>& ring)
{
    constexpr int PAIRS = Cfg<R>::PAIRS;
    constexpr int CH = Cfg<R>::CH;
    if (64 * IB < ncols) {
        int lim = ncols - 64 * IB;          // columns of this block in use (multiple of NB * CH)
        const bool top = lim <= 64;
        lim = lim > 64 ? 64 : lim;
        if (top) {
#pragma unroll
            for (int i = 0; i < NB; ++i)
                chunk_load<R, 0, IB + 1>(ring.b[i], Ut, 32 * IB + ((lim - 1 - i * CH) >> 1), -1, lane);
        }
        for (int hi = lim - 1; hi >= 0; hi -= NB * CH) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                bwd_apply<R, BT, IB>(d, ring.b[i], hi - i * CH);
                __builtin_amdgcn_sched_barrier(0);
                chunk_load<R, 0, IB + 1>(ring.b[i], Ut, 32 * IB + ((hi - (i + NB) * CH) >> 1), -1, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if constexpr (IB > 0) {
        Ring<R, 0, IB, NB> next;
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int p = 0; p < PAIRS; ++p)
#pragma unroll
                for (int k = 0; k < IB; ++k) next.b[i].v[p][k] = ring.b[i].v[p][k];
        bwd_block<R, BT, NB, IB - 1>(d, Ut, lane, ncols, next);
    }
}

// ---------------------------------------------------------------------------------------
// state -> residual prologues
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
        const int pa = a >= 0 ? T.parent[a] : 0;
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

// ---------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------
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

template <int R, int BT>
__global__ void __launch_bounds__(256) k_logpdf(MvnDev M, const double* __restrict__ X, int64_t ldx, int64_t batch,
                                                double* __restrict__ ll)
{
    constexpr int NB = Cfg<R>::NB;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t b0 = wave * BT;
    if (b0 >= batch) return;  // wave-uniform exit
    Ring<R, 0, R, NB> ring;
    fwd_prologue<R, BT, NB>(ring, M.Ft, lane);   // factor stream starts before the state arrives
    double d[R][BT];
    load_rawx<R, BT>(d, M, X, ldx, b0, batch, lane);
    fwd_block<R, BT, NB, 0>(d, M.Ft, lane, M.ncols, ring);
    finish_ll<R, BT>(d, M, b0, batch, ll, lane);
}

template <int R, int BT>
__global__ void __launch_bounds__(256) k_grad(MvnDev M, const double* __restrict__ X, int64_t ldx, int64_t batch,
                                              double* __restrict__ ll, double* __restrict__ G, int64_t ldg)
{
    constexpr int NB = Cfg<R>::NB;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t b0 = wave * BT;
    if (b0 >= batch) return;
    Ring<R, 0, R, NB> ring;
    fwd_prologue<R, BT, NB>(ring, M.Ft, lane);   // factor stream starts before the state arrives
    double d[R][BT];
    load_rawx<R, BT>(d, M, X, ldx, b0, batch, lane);
    fwd_block<R, BT, NB, 0>(d, M.Ft, lane, M.ncols, ring);
    finish_ll<R, BT>(d, M, b0, batch, ll, lane);
    // backward: y = L^-T z.  Row scaling first (z_r / L_rr), then the mirrored sweep.
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const double iv = M.invdiag[64 * k + lane];
#pragma unroll
        for (int c = 0; c < BT; ++c) d[k][c] *= iv;
    }
    {
        Ring<R, 0, R, NB> rb;
        bwd_block<R, BT, NB, R - 1>(d, M.Ut, lane, M.ncols, rb);
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int row = 64 * k + lane;
#pragma unroll
        for (int c = 0; c < BT; ++c)
            if (row < M.n && b0 + c < batch) G[(b0 + c) * ldg + row] = -d[k][c];
    }
}

template <int R, int BT>
__global__ void __launch_bounds__(256) k_tree_logpdf(MvnDev M, TreeDev T, const double* __restrict__ H,
                                                     const double* __restrict__ Rt, int64_t lds,
                                                     const double* __restrict__ tH, const double* __restrict__ rMu,
                                                     int64_t batch, double* __restrict__ ll, double* __restrict__ logjac)
{
    constexpr int NB = Cfg<R>::NB;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t b0 = wave * BT;
    if (b0 >= batch) return;
    Ring<R, 0, R, NB> ring;
    fwd_prologue<R, BT, NB>(ring, M.Ft, lane);
    double d[R][BT], dist[R][BT];
    load_tree<R, BT>(d, dist, M, T, H, Rt, lds, tH, rMu, b0, batch, lane);
    if (logjac != nullptr && lane == 0) {
#pragma unroll
        for (int c = 0; c < BT; ++c)
            if (b0 + c < batch) logjac[b0 + c] = log(1.0 / dist[0][c]);  // app/Probability.hs:394, 409
    }
    fwd_block<R, BT, NB, 0>(d, M.Ft, lane, M.ncols, ring);
    finish_ll<R, BT>(d, M, b0, batch, ll, lane);
}

// Tree gradient: chain rule from g = d ll / d distances back to heights, rates, tH, rMu
// (oracle/mvn_oracle.c orc_tree_grad_full states the same formulas).
// One wave per chain (BT = 1); e[v] = s * g[row(v)] * rate[v] is exchanged through LDS.
template <int R>
__global__ void __launch_bounds__(256) k_tree_grad(MvnDev M, TreeDev T, const double* __restrict__ H,
                                                   const double* __restrict__ Rt, int64_t lds,
                                                   const double* __restrict__ tH, const double* __restrict__ rMu,
                                                   int64_t batch, double* __restrict__ ll, double* __restrict__ gH,
                                                   double* __restrict__ gR, double* __restrict__ gtH,
                                                   double* __restrict__ grMu)
{
    extern __shared__ double smem[];       // [waves per block][n_nodes_padded]
    constexpr int NB = Cfg<R>::NB;
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= batch) return;
    double* e = smem + (size_t)wib * T.n_nodes_pad;
    Ring<R, 0, R, NB> ring;
    fwd_prologue<R, 1, NB>(ring, M.Ft, lane);
    double d[R][1], dist[R][1];
    load_tree<R, 1>(d, dist, M, T, H, Rt, lds, tH, rMu, b, batch, lane);
    fwd_block<R, 1, NB, 0>(d, M.Ft, lane, M.ncols, ring);
    finish_ll<R, 1>(d, M, b, batch, ll, lane);
#pragma unroll
    for (int k = 0; k < R; ++k) d[k][0] *= M.invdiag[64 * k + lane];
    {
        Ring<R, 0, R, NB> rb;
        bwd_block<R, 1, NB, R - 1>(d, M.Ut, lane, M.ncols, rb);
    }
    // now d = y = Sigma^-1 (dist - mu); g = -y
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
            const int pa = T.parent[a];
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int v = lane; v < T.n_nodes; v += 64) {
        double acc = (v == 0) ? 0.0 : -e[v];
        for (int ci = T.child_ptr[v]; ci < T.child_ptr[v + 1]; ++ci) acc += e[T.child_idx[ci]];
        gH[b * lds + v] = acc;
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
static inline int pick_bt(int R, int64_t batch)
{
    // Small batches: one chain per wave to put as many SIMDs as possible on the dependency
    // chain.  Large batches: two chains per wave halve the L2 -> CU traffic of the factor.
    const int64_t waves_bt1 = batch;
    (void)R;
    return (waves_bt1 > 8192) ? 2 : 1;
}

template <int R>
static hipError_t launch_logpdf_R(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st)
{
    const int bt = pick_bt(R, batch);
    const int wpb = getenv("MCD_WPB") ? atoi(getenv("MCD_WPB")) : 2;  // waves per block
    const int64_t waves = (batch + bt - 1) / bt;
    const unsigned grid = (unsigned)((waves + wpb - 1) / wpb);
    if (bt == 1)
        hipLaunchKernelGGL((k_logpdf<R, 1>), dim3(grid), dim3(64 * wpb), 0, st, M, X, ldx, batch, ll);
    else
        hipLaunchKernelGGL((k_logpdf<R, 2>), dim3(grid), dim3(64 * wpb), 0, st, M, X, ldx, batch, ll);
    return hipGetLastError();
}

template <int R>
static hipError_t launch_grad_R(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, double* G,
                                int64_t ldg, hipStream_t st)
{
    const int wpb = 2;
    const unsigned grid = (unsigned)((batch + wpb - 1) / wpb);
    hipLaunchKernelGGL((k_grad<R, 1>), dim3(grid), dim3(64 * wpb), 0, st, M, X, ldx, batch, ll, G, ldg);
    return hipGetLastError();
}

template <int R>
static hipError_t launch_tree_logpdf_R(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                                       const double* tH, const double* rMu, int64_t batch, double* ll, double* logjac,
                                       hipStream_t st)
{
    const int wpb = 2;
    const unsigned grid = (unsigned)((batch + wpb - 1) / wpb);
    hipLaunchKernelGGL((k_tree_logpdf<R, 1>), dim3(grid), dim3(64 * wpb), 0, st, M, T, H, Rt, lds, tH, rMu, batch, ll,
                       logjac);
    return hipGetLastError();
}

template <int R>
static hipError_t launch_tree_grad_R(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                                     const double* tH, const double* rMu, int64_t batch, double* ll, double* gH,
                                     double* gR, double* gtH, double* grMu, hipStream_t st)
{
    const int wpb = 2;
    const unsigned grid = (unsigned)((batch + wpb - 1) / wpb);
    const size_t shmem = sizeof(double) * (size_t)wpb * (size_t)T.n_nodes_pad;
    hipLaunchKernelGGL((k_tree_grad<R>), dim3(grid), dim3(64 * wpb), shmem, st, M, T, H, Rt, lds, tH, rMu, batch, ll,
                       gH, gR, gtH, grMu);
    return hipGetLastError();
}

#define MCD_DISPATCH_R(R_, CALL)                    \
    switch (R_) {                                   \
        case 1: return CALL(1);                     \
        case 2: return CALL(2);                     \
        case 3: return CALL(3);                     \
        case 4: return CALL(4);                     \
        case 6: return CALL(6);                     \
        case 8: return CALL(8);                     \
        case 12: return CALL(12);                   \
        case 16: return CALL(16);                   \
        default: return hipErrorInvalidValue;       \
    }

int sweep_chunk_columns(int R)
{
    // columns consumed per loop trip of the sweeps = Cfg<R>::NB * Cfg<R>::CH; the swept column
    // count is rounded up to it (extra columns are zero padding).
    const int pairs = (R <= 1) ? 4 : (R <= 4) ? 2 : 1;
    const int nb = (R <= 8) ? 8 : 4;
    return nb * 2 * pairs;
}

int padded_blocks(int n)
{
    const int r = (n + 63) / 64;
    const int allowed[] = {1, 2, 3, 4, 6, 8, 12, 16};
    for (int a : allowed)
        if (r <= a) return a;
    return -1;
}

hipError_t launch_logpdf(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
#define CALL(R) launch_logpdf_R<R>(M, X, ldx, batch, ll, st)
    MCD_DISPATCH_R(M.R, CALL)
#undef CALL
}

hipError_t launch_grad(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, double* G, int64_t ldg,
                       hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
#define CALL(R) launch_grad_R<R>(M, X, ldx, batch, ll, G, ldg, st)
    MCD_DISPATCH_R(M.R, CALL)
#undef CALL
}

hipError_t launch_tree_logpdf(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                              const double* tH, const double* rMu, int64_t batch, double* ll, double* logjac,
                              hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
#define CALL(R) launch_tree_logpdf_R<R>(M, T, H, Rt, lds, tH, rMu, batch, ll, logjac, st)
    MCD_DISPATCH_R(M.R, CALL)
#undef CALL
}

hipError_t launch_tree_grad(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                            const double* tH, const double* rMu, int64_t batch, double* ll, double* gH, double* gR,
                            double* gtH, double* grMu, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
#define CALL(R) launch_tree_grad_R<R>(M, T, H, Rt, lds, tH, rMu, batch, ll, gH, gR, gtH, grMu, st)
    MCD_DISPATCH_R(M.R, CALL)
#undef CALL
}

}  // namespace mcd
