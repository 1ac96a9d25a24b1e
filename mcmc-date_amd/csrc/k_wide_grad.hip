// k_wide_grad.hip -- log-density AND gradient of many chains at once, multiply form on the fp64 matrix cores (gfx950).
//
// Companion of k_wide.hip for N <= 256 (one staged chunk): with W = L^-1,
//     z = W (x - mu),   ll = c - 1/2 (logdet Sigma + |z|^2),   d ll / d x = -W^T z = -Sigma^-1 (x - mu)
// (the reference gets the gradient by AD of likelihoodFunctionG, app/Probability.hs:361-388).
// Both products are triangular and run as v_mfma_f64_16x16x4_f64 tiles:
//   forward   z  = W r    row block b needs the k tiles 0 .. 4 (b + 1) - 1     (tile stream MvnDev::Wt)
//   backward  y  = W^T z  row block b needs the k tiles 4 b .. 4 NB - 1         (tile stream MvnDev::Wtb)
// A wave owns the row blocks in slots w and 15 - w in both passes (equal work in each).  z goes from the accumulators
// into the LDS chunk that held the residuals (the f64 result layout has the chain on lane & 15, so the write is the
// transpose-free inverse of the B-operand read), y replaces z the same way, and the outputs leave LDS with coalesced
// stores.  For tree states the chain rule to heights, rates, tH and rMu (k_tree_grad.hip)
//     d ll/d r_v = s g t_v,   d ll/d h_v = s (sum_children g_c r_c - g_v r_v),   d ll/d tH = g.d / tH,   d ll/d rMu = g.d / rMu
// runs on the LDS copy: e[v] = s g[slot(v)] r_v is scattered into the same rows by node id, then every node gathers its
// children.
#include "wide_device.hpp"
#include <atomic>

namespace mcd {

struct WideGradOut {
    double* ll;
    double* G;          // raw x: [batch][ldg]
    int64_t ldg;
    double *gH, *gR;    // tree state: [batch][lds]
    double *gtH, *grMu; // [batch]
};

template <int CT>
__device__ __forceinline__ void store_tile_rows(double* rs, const d4 (&acc)[CT], int b, int col, int kq, double sign)
{
    // result layout of v_mfma_f64_16x16x4_f64: chain = lane & 15, row = (lane >> 4) + 4 q
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int q = 0; q < 4; ++q) rs[(ct * 16 + col) * WD_LD + 16 * b + kq + 4 * q] = sign * acc[ct][q];
    }
}

template <int CT, bool TREE>
__global__ void __launch_bounds__(64 * WD_WAVES) k_wide_grad(MvnDev M, WideSrc A, WideGradOut O, int64_t batch)
{
    extern __shared__ double smem[];
    double* rs = smem;                                   // [CT * 16][WD_LD]: residuals, then z, then g (then e by node)
    double* part = smem + CT * 16 * WD_LD;               // [WD_WAVES][CT * 16]
    double* scs = part + WD_WAVES * CT * 16;             // [CT * 16] tH * rMu per chain (tree state)
    double* gpart = scs + CT * 16;                       // [CT * 16][4] partial g.d per chain and wave quarter + [CT * 16] root branch (tree state)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b0 = (int64_t)blockIdx.x * (CT * 16);
    const int N = M.n, nb = (N + 15) >> 4;               // nb <= 16: one super block, one chunk
    const int col = lane & 15, kq = lane >> 4;
    const int shift = 16 - nb;
    const int bA = wave - shift, bB = 15 - wave - shift;  // this wave's row blocks (or < 0)

    WD_T(0);
    if constexpr (TREE) {
        if (tid < CT * 16) {
            const int64_t b = (b0 + tid < batch) ? b0 + tid : batch - 1;
            scs[tid] = A.tH[b] * A.rMu[b];
        }
    }
    __syncthreads();
    double tk[CT * 8], rk[CT * 8];                        // tree state: branch duration and rate of (slot j, chain row it)
    if constexpr (TREE) {
        wide_stage_tree_keep<CT>(rs, scs, M, A, b0, batch, tid, tk, rk);
        wide_stage_root<CT>(rs, scs, M, A, b0, batch, false, tid);
    } else {
        wide_stage<CT, false>(rs, scs, M, A, b0, batch, 0, tid);
    }
    __syncthreads();
    WD_T(1);

    // ---- forward: z = W r -------------------------------------------------------------------------------------------
    d4 accA[CT], accB[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) accA[ct] = accB[ct] = d4{0.0, 0.0, 0.0, 0.0};
    if (bA >= 0) wide_tri_pass<CT>(M.Wt + (int64_t)(2 * bA * (bA + 1)) * 64 + lane, 4 * (bA + 1), 0, rs, col, kq, accA);
    if (bB >= 0) wide_tri_pass<CT>(M.Wt + (int64_t)(2 * bB * (bB + 1)) * 64 + lane, 4 * (bB + 1), 0, rs, col, kq, accB);
    double ssq[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        ssq[ct] = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ssq[ct] = fma(accA[ct][q], accA[ct][q], ssq[ct]);
            ssq[ct] = fma(accB[ct][q], accB[ct][q], ssq[ct]);
        }
        double v = ssq[ct];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) part[wave * (CT * 16) + ct * 16 + lane] = v;
    }
    WD_T(2);
    __syncthreads();                                      // every wave has read the residuals
    if (bA >= 0) store_tile_rows<CT>(rs, accA, bA, col, kq, 1.0);
    if (bB >= 0) store_tile_rows<CT>(rs, accB, bB, col, kq, 1.0);
    if (tid < CT * 16 && b0 + tid < batch) {
        double q = 0.0;
#pragma unroll
        for (int w = 0; w < WD_WAVES; ++w) q += part[w * (CT * 16) + tid];
        O.ll[b0 + tid] = M.c + (-0.5) * (M.logdet + q);   // app/Probability.hs:169
    }
    __syncthreads();
    WD_T(3);

    // ---- backward: y = W^T z; block b starts at tile 4 (b nb - b (b - 1) / 2) of the stream, k tiles 4 b .. 4 nb - 1 ----
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) accA[ct] = accB[ct] = d4{0.0, 0.0, 0.0, 0.0};
    if (bA >= 0) wide_tri_pass<CT>(M.Wtb + (int64_t)(4 * (bA * nb - bA * (bA - 1) / 2)) * 64 + lane, 4 * (nb - bA), 4 * bA, rs, col, kq, accA);
    if (bB >= 0) wide_tri_pass<CT>(M.Wtb + (int64_t)(4 * (bB * nb - bB * (bB - 1) / 2)) * 64 + lane, 4 * (nb - bB), 4 * bB, rs, col, kq, accB);
    WD_T(4);
    __syncthreads();                                      // every wave has read z
    if (bA >= 0) store_tile_rows<CT>(rs, accA, bA, col, kq, -1.0);   // g = -y
    if (bB >= 0) store_tile_rows<CT>(rs, accB, bB, col, kq, -1.0);
    __syncthreads();
    WD_T(5);

    // ---- outputs: thread = (column j, chain rows ch0, ch0 + 2, ...) as in wide_stage ----------------------------------
    const int j = tid & (WD_SB - 1), ch0 = tid >> 8;
    const bool live = j < N;
    if constexpr (!TREE) {
#pragma unroll 4
        for (int it = 0; it < CT * 8; ++it) {
            const int ch = ch0 + 2 * it;
            const int64_t b = b0 + ch;
            if (live && b < batch) O.G[b * O.ldg + j] = rs[ch * WD_LD + j];
        }
    } else {
        const int n_nodes = A.T.n_nodes;                  // = N + 2 <= WD_LD
        const int a = live ? A.T.slot_node[j] : 0;
        const int rr = A.T.root_right;
        // children of the node(s) this thread gathers for below (requested now, used after two barriers)
        const int v0 = j, v1 = j + WD_SB;
        const int c00 = v0 < n_nodes ? A.T.child_ptr[v0] : 0, c01 = v0 < n_nodes ? A.T.child_ptr[v0 + 1] : 0;
        const int c10 = v1 < n_nodes ? A.T.child_ptr[v1] : 0, c11 = v1 < n_nodes ? A.T.child_ptr[v1 + 1] : 0;
        const int k00 = c01 - c00 > 0 ? A.T.child_idx[c00] : 0, k01 = c01 - c00 > 1 ? A.T.child_idx[c00 + 1] : 0;
        double ev[CT * 8], gd[CT * 8];
#pragma unroll
        for (int it = 0; it < CT * 8; ++it) {
            const int ch = ch0 + 2 * it;
            const int64_t b = b0 + ch;
            const double g = live ? rs[ch * WD_LD + j] : 0.0;
            const double sg = scs[ch] * g;
            ev[it] = sg * rk[it];
            if (live && b < batch) O.gR[b * A.lds + a] = sg * tk[it];          // d ll / d r_v = s g t_v
            gd[it] = g * ((tk[it] * rk[it]) * scs[ch]);
        }
        // the second root branch shares slot 0 (sumFirstTwo): one chain per thread
        double e_rr = 0.0, gd_rr = 0.0;
        if (tid < CT * 16) {
            const int64_t b = (b0 + tid < batch) ? b0 + tid : batch - 1;
            const double* h = A.H + b * A.lds;
            const double r2 = A.Rt[b * A.lds + rr];
            const double t2 = h[0] - h[rr];
            const double g0 = rs[tid * WD_LD];
            const double sg = scs[tid] * g0;
            e_rr = sg * r2;
            gd_rr = g0 * ((t2 * r2) * scs[tid]);
            if (b0 + tid < batch) {
                O.gR[b * A.lds + rr] = sg * t2;
                O.gR[b * A.lds] = 0.0;                                           // stem rate: unused by the likelihood
            }
        }
        WD_T(6);
        // g . d: the 64 columns of this wave (all chain rows at once, so the exchanges overlap), then the four waves of a
        // chain row in a fixed order below
#pragma unroll
        for (int it = 0; it < CT * 8; ++it) gd[it] = wd_wave_sum(gd[it]);
        if (lane == 0) {
#pragma unroll
            for (int it = 0; it < CT * 8; ++it) gpart[(ch0 + 2 * it) * 4 + (wave & 3)] = gd[it];
        }
        WD_T(7);
        __syncthreads();                                  // every thread holds its g-derived values: the rows can be reused
#pragma unroll
        for (int it = 0; it < CT * 8; ++it) {
            const int ch = ch0 + 2 * it;
            if (live) rs[ch * WD_LD + a] = ev[it];        // e by node id (slots -> nodes is one-to-one onto 1 .. n_nodes - 1 \ {rr})
        }
        if (tid < CT * 16) {
            rs[tid * WD_LD + rr] = e_rr;
            rs[tid * WD_LD] = 0.0;                        // the root has no branch
            gpart[CT * 16 * 4 + tid] = gd_rr;
        }
        __syncthreads();
        if (tid < CT * 16 && b0 + tid < batch) {
            const double gdot = (((gpart[tid * 4] + gpart[tid * 4 + 1]) + gpart[tid * 4 + 2]) + gpart[tid * 4 + 3]) + gpart[CT * 16 * 4 + tid];
            O.gtH[b0 + tid] = gdot / A.tH[b0 + tid];
            O.grMu[b0 + tid] = gdot / A.rMu[b0 + tid];
        }
        WD_T(8);
        // d ll / d h_v = sum_children e_c - e_v, one node per thread column (n_nodes <= 258: columns 0 .. 255 and a tail of 2)
        if (v0 < n_nodes) {
#pragma unroll 4
            for (int it = 0; it < CT * 8; ++it) {
                const int ch = ch0 + 2 * it;
                const int64_t b = b0 + ch;
                if (b >= batch) continue;
                const double* e = rs + ch * WD_LD;
                double acc = (v0 == 0) ? 0.0 : -e[v0];
                if (c01 - c00 > 0) acc += e[k00];
                if (c01 - c00 > 1) acc += e[k01];
                for (int ci = c00 + 2; ci < c01; ++ci) acc += e[A.T.child_idx[ci]];      // multifurcations
                O.gH[b * A.lds + v0] = acc;
            }
        }
        if (v1 < n_nodes) {
            for (int it = 0; it < CT * 8; ++it) {
                const int ch = ch0 + 2 * it;
                const int64_t b = b0 + ch;
                if (b >= batch) continue;
                const double* e = rs + ch * WD_LD;
                double acc = -e[v1];
                for (int ci = c10; ci < c11; ++ci) acc += e[A.T.child_idx[ci]];
                O.gH[b * A.lds + v1] = acc;
            }
        }
    }
    WD_T(9);
}

// More than 64 KiB of dynamic LDS has to be allowed once per kernel and device (a process may hold handles on several
// GPUs).  mcd_mvn_create does it for every instantiation (prepare_wide_grad), so that a first launch under stream capture needs no
// attribute call; the launchers check again.
template <int CT, bool TREE>
static hipError_t allow_lds()
{
    constexpr size_t bytes = (size_t)(CT * 16 * WD_LD + WD_WAVES * CT * 16 + CT * 16 + CT * 16 * 5) * sizeof(double);
    static std::atomic<bool> allowed[64];
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!allowed[dev].load(std::memory_order_acquire)) {
        if (hipError_t e = hipFuncSetAttribute((const void*)k_wide_grad<CT, TREE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)) return e;
        allowed[dev].store(true, std::memory_order_release);
    }
    return hipSuccess;
}

template <int CT, bool TREE>
static hipError_t launch_ct(const MvnDev& M, const WideSrc& A, const WideGradOut& O, int64_t batch, hipStream_t st)
{
    constexpr size_t bytes = (size_t)(CT * 16 * WD_LD + WD_WAVES * CT * 16 + CT * 16 + CT * 16 * 5) * sizeof(double);
    if (hipError_t e = allow_lds<CT, TREE>()) return e;
    const unsigned grid = (unsigned)((batch + CT * 16 - 1) / (CT * 16));
    hipLaunchKernelGGL((k_wide_grad<CT, TREE>), dim3(grid), dim3(64 * WD_WAVES), bytes, st, M, A, O, batch);
    return hipGetLastError();
}

template <bool TREE>
static hipError_t launch_wide(const MvnDev& M, const WideSrc& A, const WideGradOut& O, int64_t batch, hipStream_t st)
{
    if (M.Wt == nullptr || M.Wtb == nullptr || M.n > WD_SB) return hipErrorInvalidValue;
    if (wide_chain_tiles(batch) == 1) return launch_ct<1, TREE>(M, A, O, batch, st);
    return launch_ct<2, TREE>(M, A, O, batch, st);       // (64 chains per workgroup would not leave room for the outputs' registers)
}

hipError_t launch_grad_wide(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, double* G, int64_t ldg,
                            hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    WideSrc A{};
    A.X = X;
    A.ldx = ldx;
    WideGradOut O{};
    O.ll = ll;
    O.G = G;
    O.ldg = ldg;
    return launch_wide<false>(M, A, O, batch, st);
}

hipError_t launch_tree_grad_wide(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                 const double* rMu, int64_t batch, double* ll, double* gH, double* gR, double* gtH, double* grMu,
                                 hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (T.n_nodes > WD_LD) return hipErrorInvalidValue;
    WideSrc A{};
    A.T = T;
    A.H = H;
    A.Rt = Rt;
    A.lds = lds;
    A.tH = tH;
    A.rMu = rMu;
    WideGradOut O{};
    O.ll = ll;
    O.gH = gH;
    O.gR = gR;
    O.gtH = gtH;
    O.grMu = grMu;
    return launch_wide<true>(M, A, O, batch, st);
}

hipError_t prepare_wide_grad()
{
    if (hipError_t e = allow_lds<1, false>()) return e;
    if (hipError_t e = allow_lds<1, true>()) return e;
    if (hipError_t e = allow_lds<2, false>()) return e;
    if (hipError_t e = allow_lds<2, true>()) return e;
    return hipSuccess;
}

}  // namespace mcd

#ifdef MCD_WIDE_STAMP_GRAD
extern "C" int mcd_wide_debug_stamps(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mcd::g_wide_dbg), sizeof(unsigned long long) * mcd::WD_WAVES * 16);
}
#endif
