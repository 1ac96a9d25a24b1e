// k_wide_grad_mc.hip -- log-density and gradient in the multiply form for 256 < N <= 1024 (several staged chunks).
//
// Same two triangular products as k_wide_grad.hip (z = W r, y = W^T z, d ll / d x = -y), but z and y no longer fit the
// one LDS chunk of a workgroup: they travel through the caller's own output buffer, which is dead until the very end --
// the gradient array G for raw x, the height-gradient array gH for tree states (both at least N doubles per chain):
//   forward   super block s (256 rows): chunks 0 .. s of the residuals staged from the inputs, z rows -> buffer;
//   backward  super block s: chunks s .. NS - 1 of z staged from the buffer, g = -y rows -> buffer rows of chunk s (that
//             chunk of z is not needed by any later super block);
//   raw x:    the buffer now holds the gradient;
//   tree:     the chain rule (k_tree_grad.hip) in sub-batches of 4 CT chains: g by slot is read from the
//             buffer, d ll / d r_v goes to gR, e_v = s g r_v is scattered by node id into LDS (the chunk buffer, 1032
//             doubles per chain), every node gathers its children and overwrites the buffer row with d ll / d h_v.
// Only the waves of one workgroup ever touch a chain's rows, ordered by workgroup barriers.
#include "wide_device.hpp"
#include <atomic>

namespace mcd {

constexpr int WD_EL = 1032;      // LDS row stride of the e-by-node rows (n_nodes <= 1026)

struct WideGradMcOut {
    double* ll;
    double* buf;        // G (raw x) or gH (tree): [batch][ldb], scratch for z and g, final output
    int64_t ldb;
    double* gR;         // tree state: [batch][ldb]
    double *gtH, *grMu; // [batch]
};

template <int CT>
__device__ __forceinline__ void store_tile_global(double* buf, int64_t ldb, int64_t b0, int64_t batch, int n, const d4 (&acc)[CT], int64_t ib,
                                                  int col, int kq, double sign)
{
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int64_t b = b0 + ct * 16 + col;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row = 16 * ib + kq + 4 * q;
            if (b < batch && row < n) buf[b * ldb + row] = sign * acc[ct][q];
        }
    }
}

// one 256-column chunk of the buffer into LDS (zeros beyond N and beyond the batch)
template <int CT>
__device__ __forceinline__ void stage_buffer_chunk(double* rs, const double* buf, int64_t ldb, int64_t b0, int64_t batch, int n, int kc0, int tid)
{
    const int j = tid & (WD_SB - 1), k = kc0 + j, ch0 = tid >> 8;
    double v[8];
#pragma unroll 1
    for (int g = 0; g < CT * 8; g += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t b = b0 + ch0 + 2 * (g + i);
            v[i] = (k < n && b < batch) ? buf[b * ldb + k] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) rs[(ch0 + 2 * (g + i)) * WD_LD + j] = v[i];
    }
}

template <int CT, bool TREE>
__global__ void __launch_bounds__(64 * WD_WAVES) k_wide_grad_mc(MvnDev M, WideSrc A, WideGradMcOut O, int64_t batch)
{
    extern __shared__ double smem[];
    double* rs = smem;                                   // [CT * 16][WD_LD] chunk buffer; later [4 CT][WD_EL] e by node
    double* part = smem + CT * 16 * WD_LD;               // [WD_WAVES][CT * 16]
    double* scs = part + WD_WAVES * CT * 16;             // [CT * 16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b0 = (int64_t)blockIdx.x * (CT * 16);
    const int N = M.n, NB = (N + 15) >> 4, NS = (NB + 15) >> 4;
    const int col = lane & 15, kq = lane >> 4;

    if constexpr (TREE) {
        if (tid < CT * 16) {
            const int64_t b = (b0 + tid < batch) ? b0 + tid : batch - 1;
            scs[tid] = A.tH[b] * A.rMu[b];
        }
    }
    double ssq[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) ssq[ct] = 0.0;

    // ---- forward: z = W r, super block by super block ----------------------------------------------------------------
    for (int s = 0; s < NS; ++s) {
        const int nb = (NB - 16 * s < 16) ? NB - 16 * s : 16, shift = 16 - nb;
        const int bA = wave - shift, bB = 15 - wave - shift;
        const int64_t ibA = 16 * s + bA, ibB = 16 * s + bB;
        d4 accA[CT], accB[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) accA[ct] = accB[ct] = d4{0.0, 0.0, 0.0, 0.0};
        for (int c = 0; c <= s; ++c) {
            __syncthreads();
            wide_stage<CT, TREE>(rs, scs, M, A, b0, batch, c * WD_SB, tid);
            if constexpr (TREE) {
                if (c == 0) wide_stage_root<CT>(rs, scs, M, A, b0, batch, false, tid);
            }
            __syncthreads();
            const int ntA = bA >= 0 ? (c < s ? WD_SB / 4 : 4 * (bA + 1)) : 0;
            const int ntB = bB >= 0 ? (c < s ? WD_SB / 4 : 4 * (bB + 1)) : 0;
            wide_tri_pass<CT>(M.Wt + ((bA >= 0 ? 2 * ibA * (ibA + 1) : 0) + (WD_SB / 4) * c) * 64 + lane, ntA, 0, rs, col, kq, accA);
            wide_tri_pass<CT>(M.Wt + ((bB >= 0 ? 2 * ibB * (ibB + 1) : 0) + (WD_SB / 4) * c) * 64 + lane, ntB, 0, rs, col, kq, accB);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ssq[ct] = fma(accA[ct][q], accA[ct][q], ssq[ct]);
                ssq[ct] = fma(accB[ct][q], accB[ct][q], ssq[ct]);
            }
        }
        if (bA >= 0) store_tile_global<CT>(O.buf, O.ldb, b0, batch, N, accA, ibA, col, kq, 1.0);
        if (bB >= 0) store_tile_global<CT>(O.buf, O.ldb, b0, batch, N, accB, ibB, col, kq, 1.0);
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        double v = ssq[ct];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) part[wave * (CT * 16) + ct * 16 + lane] = v;
    }
    __syncthreads();                                      // all of z is in the buffer; the partial sums are in LDS
    if (tid < CT * 16 && b0 + tid < batch) {
        double q = 0.0;
#pragma unroll
        for (int w = 0; w < WD_WAVES; ++w) q += part[w * (CT * 16) + tid];
        O.ll[b0 + tid] = M.c + (-0.5) * (M.logdet + q);   // app/Probability.hs:169
    }

    // ---- backward: y = W^T z; block ib = k tiles 4 ib .. 4 NB - 1, stream offset 4 (ib NB - ib (ib - 1) / 2) -----------
    for (int s = 0; s < NS; ++s) {
        const int nb = (NB - 16 * s < 16) ? NB - 16 * s : 16, shift = 16 - nb;
        const int bA = wave - shift, bB = 15 - wave - shift;
        const int64_t ibA = 16 * s + bA, ibB = 16 * s + bB;
        d4 accA[CT], accB[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) accA[ct] = accB[ct] = d4{0.0, 0.0, 0.0, 0.0};
        for (int c = s; c < NS; ++c) {
            __syncthreads();
            stage_buffer_chunk<CT>(rs, O.buf, O.ldb, b0, batch, N, c * WD_SB, tid);
            __syncthreads();
            const int hi = (4 * NB < (WD_SB / 4) * (c + 1)) ? 4 * NB : (WD_SB / 4) * (c + 1);
            if (bA >= 0) {
                const int lo = (4 * (int)ibA > (WD_SB / 4) * c) ? 4 * (int)ibA : (WD_SB / 4) * c;
                wide_tri_pass<CT>(M.Wtb + (4 * (ibA * NB - ibA * (ibA - 1) / 2) + (lo - 4 * ibA)) * 64 + lane, hi - lo, lo - (WD_SB / 4) * c, rs, col, kq, accA);
            }
            if (bB >= 0) {
                const int lo = (4 * (int)ibB > (WD_SB / 4) * c) ? 4 * (int)ibB : (WD_SB / 4) * c;
                wide_tri_pass<CT>(M.Wtb + (4 * (ibB * NB - ibB * (ibB - 1) / 2) + (lo - 4 * ibB)) * 64 + lane, hi - lo, lo - (WD_SB / 4) * c, rs, col, kq, accB);
            }
        }
        // chunk s of z has been staged for the last time (by this super block, before its first barrier pair ended)
        if (bA >= 0) store_tile_global<CT>(O.buf, O.ldb, b0, batch, N, accA, ibA, col, kq, -1.0);   // g = -y
        if (bB >= 0) store_tile_global<CT>(O.buf, O.ldb, b0, batch, N, accB, ibB, col, kq, -1.0);
    }
    if constexpr (!TREE) return;

    // ---- tree states: chain rule, 4 CT chains at a time -----------------------------------------------------------------
    if constexpr (TREE) {
        constexpr int SBC = CT * 4;                        // chains per sub-batch: SBC * WD_EL doubles <= the chunk buffer
        const int n_nodes = A.T.n_nodes, rr = A.T.root_right;
        double* eb = rs;
        double* gpart = part;                              // [WD_WAVES][SBC] (the ll partial sums have been consumed)
        for (int sb = 0; sb < CT * 16 / SBC; ++sb) {
            __syncthreads();                               // g rows complete (first pass) / previous sub-batch done with eb
            const int c0 = sb * SBC;
            double gd[SBC];
#pragma unroll
            for (int i = 0; i < SBC; ++i) gd[i] = 0.0;
            for (int j = tid; j < N; j += 64 * WD_WAVES) {
                const int a = A.T.slot_node[j], pa = A.T.slot_parent[j];
#pragma unroll
                for (int i = 0; i < SBC; ++i) {
                    const int64_t b = b0 + c0 + i;
                    if (b >= batch) continue;
                    const double g = O.buf[b * O.ldb + j];
                    const double t = A.H[b * A.lds + pa] - A.H[b * A.lds + a];
                    const double ra = A.Rt[b * A.lds + a];
                    const double sg = scs[c0 + i] * g;
                    O.gR[b * A.lds + a] = sg * t;                              // d ll / d r_v = s g t_v
                    eb[i * WD_EL + a] = sg * ra;
                    gd[i] += g * ((t * ra) * scs[c0 + i]);
                }
            }
            double e_rr = 0.0, gd_rr = 0.0;
            if (tid < SBC && b0 + c0 + tid < batch) {      // the second root branch shares slot 0; the root has no branch
                const int64_t b = b0 + c0 + tid;
                const double r2 = A.Rt[b * A.lds + rr];
                const double t2 = A.H[b * A.lds] - A.H[b * A.lds + rr];
                const double g0 = O.buf[b * O.ldb];
                const double sg = scs[c0 + tid] * g0;
                e_rr = sg * r2;
                gd_rr = g0 * ((t2 * r2) * scs[c0 + tid]);
                O.gR[b * A.lds + rr] = sg * t2;
                O.gR[b * A.lds] = 0.0;                     // stem rate: unused by the likelihood
                eb[tid * WD_EL + rr] = e_rr;
                eb[tid * WD_EL] = 0.0;
            }
#pragma unroll
            for (int i = 0; i < SBC; ++i) {
                const double v = wd_wave_sum(gd[i]);
                if (lane == 0) gpart[wave * SBC + i] = v;
            }
            __syncthreads();                               // e rows complete, every g of these chains has been read
            if (tid < SBC && b0 + c0 + tid < batch) {
                double gdot = gd_rr;
#pragma unroll
                for (int w = 0; w < WD_WAVES; ++w) gdot += gpart[w * SBC + tid];
                O.gtH[b0 + c0 + tid] = gdot / A.tH[b0 + c0 + tid];
                O.grMu[b0 + c0 + tid] = gdot / A.rMu[b0 + c0 + tid];
            }
            for (int v = tid; v < n_nodes; v += 64 * WD_WAVES) {
                const int p0 = A.T.child_ptr[v], p1 = A.T.child_ptr[v + 1];
#pragma unroll
                for (int i = 0; i < SBC; ++i) {
                    const int64_t b = b0 + c0 + i;
                    if (b >= batch) continue;
                    const double* e = eb + i * WD_EL;
                    double acc = (v == 0) ? 0.0 : -e[v];
                    for (int ci = p0; ci < p1; ++ci) acc += e[A.T.child_idx[ci]];
                    O.buf[b * O.ldb + v] = acc;            // d ll / d h_v replaces g in the buffer
                }
            }
        }
    }
}

// More than 64 KiB of dynamic LDS has to be allowed once per kernel and device (a process may hold handles on several
// GPUs).  mcd_mvn_create does it for every instantiation (prepare_wide_grad_mc), so that a first launch under stream capture needs no
// attribute call; the launchers check again.
template <int CT, bool TREE>
static hipError_t allow_lds()
{
    constexpr size_t bytes = (size_t)(CT * 16 * WD_LD + WD_WAVES * CT * 16 + CT * 16) * sizeof(double);
    static std::atomic<bool> allowed[64];
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!allowed[dev].load(std::memory_order_acquire)) {
        if (hipError_t e = hipFuncSetAttribute((const void*)k_wide_grad_mc<CT, TREE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)) return e;
        allowed[dev].store(true, std::memory_order_release);
    }
    return hipSuccess;
}

template <int CT, bool TREE>
static hipError_t launch_ct(const MvnDev& M, const WideSrc& A, const WideGradMcOut& O, int64_t batch, hipStream_t st)
{
    constexpr size_t bytes = (size_t)(CT * 16 * WD_LD + WD_WAVES * CT * 16 + CT * 16) * sizeof(double);
    static_assert(CT * 4 * WD_EL <= CT * 16 * WD_LD, "e-by-node rows must fit the chunk buffer");
    if (hipError_t e = allow_lds<CT, TREE>()) return e;
    const unsigned grid = (unsigned)((batch + CT * 16 - 1) / (CT * 16));
    hipLaunchKernelGGL((k_wide_grad_mc<CT, TREE>), dim3(grid), dim3(64 * WD_WAVES), bytes, st, M, A, O, batch);
    return hipGetLastError();
}

template <bool TREE>
static hipError_t launch_mc(const MvnDev& M, const WideSrc& A, const WideGradMcOut& O, int64_t batch, hipStream_t st)
{
    if (M.Wt == nullptr || M.Wtb == nullptr || O.ldb < M.n) return hipErrorInvalidValue;
    if (wide_chain_tiles(batch) == 1) return launch_ct<1, TREE>(M, A, O, batch, st);
    return launch_ct<2, TREE>(M, A, O, batch, st);
}

hipError_t launch_grad_wide_mc(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, double* G, int64_t ldg, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    WideSrc A{};
    A.X = X;
    A.ldx = ldx;
    WideGradMcOut O{};
    O.ll = ll;
    O.buf = G;
    O.ldb = ldg;
    return launch_mc<false>(M, A, O, batch, st);
}

hipError_t launch_tree_grad_wide_mc(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                    const double* rMu, int64_t batch, double* ll, double* gH, double* gR, double* gtH, double* grMu,
                                    hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (T.n_nodes > WD_EL) return hipErrorInvalidValue;
    WideSrc A{};
    A.T = T;
    A.H = H;
    A.Rt = Rt;
    A.lds = lds;
    A.tH = tH;
    A.rMu = rMu;
    WideGradMcOut O{};
    O.ll = ll;
    O.buf = gH;
    O.ldb = lds;
    O.gR = gR;
    O.gtH = gtH;
    O.grMu = grMu;
    return launch_mc<true>(M, A, O, batch, st);
}

hipError_t prepare_wide_grad_mc()
{
    if (hipError_t e = allow_lds<1, false>()) return e;
    if (hipError_t e = allow_lds<1, true>()) return e;
    if (hipError_t e = allow_lds<2, false>()) return e;
    if (hipError_t e = allow_lds<2, true>()) return e;
    return hipSuccess;
}

}  // namespace mcd
