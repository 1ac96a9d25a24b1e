// k_sparse.hip -- the sparse form of the log-density on the device (gfx950): the precision matrix as it is, CSR, no densification.
//
// Reference: logDensitySparseMultivariateNormal, app/Probability.hs:178-184 --
//     ll = c - 1/2 (log det Sigma + dx . (P !#> dx)),   dx = x - mu,   c = -N ln sqrt(2 pi),
// P = the graphical-lasso estimate of Sigma^-1 stored as an association list (app/Main.hs:142-155, 257-277; `.data` tag
// SparseS) and multiplied by hmatrix's CSR mat-vec.  It is the reference's route for trees with thousands of branches
// (tutorial/main/tutorial.org:487-496), beyond the dense kernels' N <= 1024.  State -> distances as everywhere
// (likelihoodFunctionWrapper, :195-207); d ll / d x = -P dx falls out of the same product.
//
// Mapping.  A workgroup (4 waves) owns a tile of C chains (16 .. 1: as many as fit the CU's LDS with the whole dx tile,
// xs[N][C + 1] doubles, the + 1 keeps a column's lanes on different banks).  Phase 1 stages the tile: chain vectors read
// coalesced along j (tree states: distances computed from heights and rates on the way).  Phase 2 deals the rows to the waves
// (row i to wave i mod 4); a wave walks row i's nonzeros 64 / C at a time -- lane = (nonzero slot s, chain c): the column
// index and value of a nonzero are one coalesced read shared by C lanes, its dx a 128-byte LDS segment --, folds the slots with
// lane exchanges (y_i for C chains), adds dx_i y_i to the chains' quadratic forms and, for the gradient, stores -y_i.
// Phase 3 adds the four waves' partial forms in a fixed order.  Everything is summed in a fixed order: bit-reproducible.
// HBM-bound on the CSR stream: 12 bytes per nonzero per tile from L2 (the matrix is shared by all tiles) for 2 C flops.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mvn_kernels.h"

namespace mcd {

template <int C, bool TREE, bool GRAD>
__global__ __launch_bounds__(256) void k_sparse(SparseDev S, SparseTreeDev T, const double* __restrict__ X, const double* __restrict__ Rt,
                                               int64_t ld, const double* __restrict__ tH, const double* __restrict__ rMu, int64_t batch,
                                               double* __restrict__ ll, double* __restrict__ logjac, double* __restrict__ G, int64_t ldg)
{
    extern __shared__ double xs[];                           // [n][C + 1] dx of the tile, then [4][C] partial quadratic forms
    constexpr int XS = C + 1;
    constexpr int NS = 64 / C;                               // nonzeros a wave takes per step
    const int n = S.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b0 = (int64_t)blockIdx.x * C;
    double* qred = xs + (size_t)n * XS;

    // ---- phase 1: the tile's dx into LDS, chain after chain, coalesced along j
#pragma unroll 1
    for (int c = 0; c < C; ++c) {
        const int64_t b = b0 + c;
        const bool in = b < batch;
        const int64_t bb = in ? b : batch - 1;               // a chain beyond the batch repeats the last one and stores nothing
        if constexpr (TREE) {
            const double* H = X + bb * ld;
            const double* R = Rt + bb * ld;
            const double s = tH[bb] * rMu[bb];
            for (int j = tid; j < n; j += 256) {
                const int a = T.slot_node[j], pa = T.slot_parent[j];
                double v = (H[pa] - H[a]) * R[a];            // heightTreeToLengthTree, times * rates
                if (j == 0) {
                    v = v + (H[0] - H[T.root_right]) * R[T.root_right];   // sumFirstTwo
                    v = v * s;
                    if (in && logjac) logjac[b] = log(1.0 / v);            // jacobianRootBranch, :393-410
                } else {
                    v = v * s;
                }
                xs[(size_t)j * XS + c] = v - S.mu[j];
            }
        } else {
            const double* x = X + bb * ld;
            for (int j = tid; j < n; j += 256) xs[(size_t)j * XS + c] = x[j] - S.mu[j];
        }
    }
    __syncthreads();

    // ---- phase 2: y = P dx row by row, q += dx_i y_i
    const int s = lane / C, c = lane - s * C;
    double q = 0.0;
    for (int i = wave; i < n; i += 4) {
        const int p1 = S.rowptr[i + 1];
        double acc = 0.0;
        for (int p = S.rowptr[i] + s; p < p1; p += NS) acc = fma(S.val[p], xs[(size_t)S.col[p] * XS + c], acc);
#pragma unroll
        for (int off = C; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
        q = fma(xs[(size_t)i * XS + c], acc, q);
        if constexpr (GRAD) {
            if (s == 0 && b0 + c < batch) G[(b0 + c) * ldg + i] = -acc;      // d ll / d x_i = -(P dx)_i
        }
    }
    if (s == 0) qred[wave * C + c] = q;
    __syncthreads();
    if (tid < C && b0 + tid < batch) {
        const double qq = ((qred[tid] + qred[C + tid]) + qred[2 * C + tid]) + qred[3 * C + tid];
        ll[b0 + tid] = S.c + (-0.5) * (S.logdet + qq);       // :180 (c - 1/2 (logdet + q))
    }
}

size_t sparse_lds_bytes(int n, int C) { return ((size_t)n * (C + 1) + 4 * (size_t)C) * sizeof(double); }

// chains per tile: the most that fit 150 KiB of LDS with the dx tile
int sparse_tile_chains(int n)
{
    const int cs[] = {16, 8, 4, 2, 1};
    for (int c : cs)
        if (sparse_lds_bytes(n, c) <= 150 * 1024) return c;
    return 0;
}

template <int C, bool TREE, bool GRAD>
static hipError_t launch_c(const SparseDev& S, const SparseTreeDev& T, const double* X, const double* Rt, int64_t ld, const double* tH,
                           const double* rMu, int64_t batch, double* ll, double* logjac, double* G, int64_t ldg, hipStream_t st)
{
    const size_t lds = sparse_lds_bytes(S.n, C);
    if (lds > 64 * 1024)
        if (hipError_t e = hipFuncSetAttribute((const void*)k_sparse<C, TREE, GRAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) return e;
    hipLaunchKernelGGL((k_sparse<C, TREE, GRAD>), dim3((unsigned)((batch + C - 1) / C)), dim3(256), lds, st, S, T, X, Rt, ld, tH, rMu, batch, ll,
                       logjac, G, ldg);
    return hipGetLastError();
}

template <bool TREE, bool GRAD>
static hipError_t launch_any(const SparseDev& S, const SparseTreeDev& T, const double* X, const double* Rt, int64_t ld, const double* tH,
                             const double* rMu, int64_t batch, double* ll, double* logjac, double* G, int64_t ldg, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    switch (sparse_tile_chains(S.n)) {
    case 16: return launch_c<16, TREE, GRAD>(S, T, X, Rt, ld, tH, rMu, batch, ll, logjac, G, ldg, st);
    case 8: return launch_c<8, TREE, GRAD>(S, T, X, Rt, ld, tH, rMu, batch, ll, logjac, G, ldg, st);
    case 4: return launch_c<4, TREE, GRAD>(S, T, X, Rt, ld, tH, rMu, batch, ll, logjac, G, ldg, st);
    case 2: return launch_c<2, TREE, GRAD>(S, T, X, Rt, ld, tH, rMu, batch, ll, logjac, G, ldg, st);
    case 1: return launch_c<1, TREE, GRAD>(S, T, X, Rt, ld, tH, rMu, batch, ll, logjac, G, ldg, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_sparse_logpdf(const SparseDev& S, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st)
{
    return launch_any<false, false>(S, SparseTreeDev{}, X, nullptr, ldx, nullptr, nullptr, batch, ll, nullptr, nullptr, 0, st);
}

hipError_t launch_sparse_grad(const SparseDev& S, const double* X, int64_t ldx, int64_t batch, double* ll, double* G, int64_t ldg, hipStream_t st)
{
    return launch_any<false, true>(S, SparseTreeDev{}, X, nullptr, ldx, nullptr, nullptr, batch, ll, nullptr, G, ldg, st);
}

hipError_t launch_sparse_tree_logpdf(const SparseDev& S, const SparseTreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                     const double* rMu, int64_t batch, double* ll, double* logjac, hipStream_t st)
{
    return launch_any<true, false>(S, T, H, Rt, lds, tH, rMu, batch, ll, logjac, nullptr, 0, st);
}

}  // namespace mcd
