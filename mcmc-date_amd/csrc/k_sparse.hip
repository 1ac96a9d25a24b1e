// k_sparse.hip -- the sparse form of the log-density on the device (gfx950): the precision matrix as it is, no densification.
//
// Reference: logDensitySparseMultivariateNormal, app/Probability.hs:178-184 --
//     ll = c - 1/2 (log det Sigma + dx . (P !#> dx)),   dx = x - mu,   c = -N ln sqrt(2 pi),
// P = the graphical-lasso estimate of Sigma^-1 stored as an association list (app/Main.hs:142-155, 257-277; `.data` tag
// SparseS) and multiplied by hmatrix's CSR mat-vec.  It is the reference's route for trees with thousands of branches
// (tutorial/main/tutorial.org:487-496), beyond the dense kernels' N <= 1024.  State -> distances as everywhere
// (likelihoodFunctionWrapper, :195-207); d ll / d x = -P dx falls out of the same product.
//
// Mapping: lanes = chains.  Three launches on the caller's stream:
//   k_sparse_stage   dx of every chain, TRANSPOSED to [N][Bp] (chain-minor) through 64 x 64 LDS tiles: chain vectors are read
//                    coalesced along j (tree states: the distances are formed on the way, heights and rates gathered from the chain's
//                    own row), written coalesced along the chains;
//   k_sparse_rows    a wave owns 64 chains and a chunk of RPC consecutive rows.  The chunk's nonzeros are contiguous in the
//                    row-sorted triplets: lane l loads nonzero p + l (row, column, value: coalesced), then the wave walks the 64 of
//                    them -- v_readlane puts row, column and value into SGPRs, the dx of column j for the wave's 64 chains is ONE
//                    coalesced 512-byte load from L2, the value is the scalar operand of v_fma_f64 (the sweep's broadcast trick).
//                    At a row's end y_i (64 chains) is complete: dx_i y_i joins the chunk's partial quadratic form, -y_i is stored
//                    coalesced (gradient).  Loads of 16 nonzeros are in flight before their FMAs; the FMAs run in nonzero order;
//   k_sparse_finish  ll[b] = c - 1/2 (logdet + sum over the chunks' partial forms, in chunk order); the gradient transposed back
//                    to chain-major through LDS tiles.
// Everything is summed in a fixed order: bit-reproducible.  Bound: L2 bandwidth -- one 512-byte line of dx per nonzero and 64
// chains (the matrix itself is 16 bytes per nonzero per 64 chains); no LDS capacity limit, any N.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "mvn_kernels.h"
#include "options.h"

namespace mcd {

__device__ __forceinline__ double sp_readlane64(double v, int l)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}

// dx[b][j] -> xsT[j][b] (b < Bp = batch rounded up to 64; chains beyond the batch get 0).  Block = 64 chains x 64 columns.
template <bool TREE>
__global__ __launch_bounds__(256) void k_sparse_stage(SparseDev S, SparseTreeDev T, const double* __restrict__ X, const double* __restrict__ Rt,
                                                     int64_t ld, const double* __restrict__ tH, const double* __restrict__ rMu, int64_t batch,
                                                     int64_t Bp, double* __restrict__ xsT, double* __restrict__ logjac)
{
    __shared__ double tile[64][65];
    const int n = S.n;
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    const int j0 = blockIdx.y * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int cb = w; cb < 64; cb += 4) {                     // chain b0 + cb, column j0 + lane
        const int64_t b = b0 + cb;
        const int j = j0 + lane;
        double v = 0.0;
        if (b < batch && j < n) {
            if constexpr (TREE) {
                const double* H = X + b * ld;
                const double* R = Rt + b * ld;
                const int a = T.slot_node[j], pa = T.slot_parent[j];
                double d = (H[pa] - H[a]) * R[a];            // heightTreeToLengthTree, times * rates
                if (j == 0) d = d + (H[0] - H[T.root_right]) * R[T.root_right];   // sumFirstTwo
                d = d * (tH[b] * rMu[b]);
                if (j == 0 && logjac) logjac[b] = log(1.0 / d);                    // jacobianRootBranch, :393-410
                v = d - S.mu[j];
            } else {
                v = X[b * ld + j] - S.mu[j];
            }
        }
        tile[cb][lane] = v;
    }
    __syncthreads();
    for (int jj = w; jj < 64; jj += 4) {
        const int j = j0 + jj;
        if (j < n) xsT[(int64_t)j * Bp + b0 + lane] = tile[lane][jj];
    }
}

// nonzero triplets sorted by (row, column): trow / tcol / tval [nnz]; rowptr [n + 1]
template <int RPC, bool GRAD>
__global__ __launch_bounds__(256) void k_sparse_rows(SparseDev S, int64_t Bp, const double* __restrict__ xsT, double* __restrict__ qpart,
                                                    double* __restrict__ yT)
{
    constexpr int U = 16;
    const int n = S.n;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = blockIdx.y * 4 + wave;
    const int r0 = chunk * RPC;
    if (r0 >= n) return;
    const int r1 = (r0 + RPC < n) ? r0 + RPC : n;
    const int64_t b = (int64_t)blockIdx.x * 64 + lane;
    const int p_begin = S.rowptr[r0], p_end = S.rowptr[r1];
    double q = 0.0, acc = 0.0;
    int cur = r0;                                            // row being accumulated (wave-uniform)
    double xcur = xsT[(int64_t)r0 * Bp + b];                 // its own dx (the dx_i of dx_i y_i): requested when the row begins
    auto flush_to = [&](int row) {                           // rows cur .. row - 1 are complete (rows without a nonzero: y = 0)
        while (cur < row) {
            q = fma(xcur, acc, q);
            if constexpr (GRAD) yT[(int64_t)cur * Bp + b] = -acc;    // d ll / d x_i = -(P dx)_i
            acc = 0.0;
            ++cur;
            if (cur < r1) xcur = xsT[(int64_t)cur * Bp + b];
        }
    };
    for (int p = p_begin; p < p_end; p += 64) {
        const int mine = (p + lane < p_end) ? p + lane : p_end - 1;
        const int tr = S.trow[mine], tc = S.col[mine];
        const double tv = (p + lane < p_end) ? S.val[mine] : 0.0;
        const int cnt = (p_end - p < 64) ? p_end - p : 64;
        for (int u0 = 0; u0 < cnt; u0 += U) {
            double x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int l = (u0 + u < cnt) ? u0 + u : cnt - 1;                  // (clamped: the weight below is 0 there)
                const int c = __builtin_amdgcn_readlane(tc, l);
                x[u] = xsT[(int64_t)c * Bp + b];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u0 + u < cnt) {                                              // wave-uniform
                    const int l = u0 + u;
                    const int r = __builtin_amdgcn_readlane(tr, l);
                    if (r != cur) flush_to(r);
                    acc = fma(sp_readlane64(tv, l), x[u], acc);
                }
            }
        }
    }
    flush_to(r1);
    qpart[(int64_t)chunk * Bp + b] = q;
}

// ll and, for the gradient, yT [N][Bp] -> G [batch][ldg]
template <bool GRAD>
__global__ __launch_bounds__(256) void k_sparse_finish(SparseDev S, int n_chunks, int64_t Bp, int64_t batch, const double* __restrict__ qpart,
                                                      double* __restrict__ ll, const double* __restrict__ yT, double* __restrict__ G, int64_t ldg)
{
    __shared__ double tile[64][65];
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (blockIdx.y == 0) {
        // wave w adds the chunks w, w + 4, ... (eight loads in flight), wave 0 the four partial sums in wave order: a fixed order
        const int64_t b = b0 + lane;
        double q = 0.0;
        int c = w;
        for (; c + 28 < n_chunks; c += 32) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = qpart[(int64_t)(c + 4 * u) * Bp + b];
#pragma unroll
            for (int u = 0; u < 8; ++u) q += v[u];
        }
        for (; c < n_chunks; c += 4) q += qpart[(int64_t)c * Bp + b];
        tile[w][lane] = q;
        __syncthreads();
        if (w == 0 && b < batch) {
            const double qq = ((tile[0][lane] + tile[1][lane]) + tile[2][lane]) + tile[3][lane];
            ll[b] = S.c + (-0.5) * (S.logdet + qq);                      // :180 (c - 1/2 (logdet + q))
        }
        __syncthreads();
    }
    if constexpr (GRAD) {
        const int j0 = blockIdx.y * 64;
        for (int jj = w; jj < 64; jj += 4) tile[jj][lane] = (j0 + jj < S.n) ? yT[(int64_t)(j0 + jj) * Bp + b0 + lane] : 0.0;
        __syncthreads();
        for (int cb = w; cb < 64; cb += 4) {
            const int64_t b = b0 + cb;
            if (b < batch && j0 + lane < S.n) G[b * ldg + j0 + lane] = tile[lane][cb];
        }
    }
}

// doubles of scratch a launch needs: xsT [n][Bp], qpart [chunks][Bp], yT [n][Bp] (gradient)
int sparse_rows_per_chunk(int n, int64_t batch) { return ((batch + 63) / 64) * ((n + 15) / 16) >= 512 ? 16 : 4; }
size_t sparse_scratch_doubles(int n, int64_t batch, bool grad)
{
    const size_t Bp = (size_t)((batch + 63) / 64) * 64;
    const int rpc = sparse_rows_per_chunk(n, batch);
    const size_t chunks = ((size_t)(n + rpc - 1) / rpc + 3) / 4 * 4;
    return Bp * ((size_t)n * (grad ? 2 : 1) + chunks);
}

template <bool TREE, bool GRAD>
static hipError_t launch_any(const SparseDev& S, const SparseTreeDev& T, const double* X, const double* Rt, int64_t ld, const double* tH,
                             const double* rMu, int64_t batch, double* ll, double* logjac, double* G, int64_t ldg, double* scratch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    const int n = S.n;
    const int64_t Bp = ((batch + 63) / 64) * 64;
    const int rpc = sparse_rows_per_chunk(n, batch);
    const int chunks = (n + rpc - 1) / rpc, chunk_blocks = (chunks + 3) / 4;
    double* xsT = scratch;
    double* qpart = xsT + (size_t)n * Bp;
    double* yT = qpart + (size_t)chunk_blocks * 4 * Bp;
    const unsigned gb = (unsigned)(Bp / 64), gj = (unsigned)((n + 63) / 64);
    hipLaunchKernelGGL((k_sparse_stage<TREE>), dim3(gb, gj), dim3(256), 0, st, S, T, X, Rt, ld, tH, rMu, batch, Bp, xsT, logjac);
    if (rpc == 16)
        hipLaunchKernelGGL((k_sparse_rows<16, GRAD>), dim3(gb, (unsigned)chunk_blocks), dim3(256), 0, st, S, Bp, xsT, qpart, yT);
    else
        hipLaunchKernelGGL((k_sparse_rows<4, GRAD>), dim3(gb, (unsigned)chunk_blocks), dim3(256), 0, st, S, Bp, xsT, qpart, yT);
    hipLaunchKernelGGL((k_sparse_finish<GRAD>), dim3(gb, GRAD ? gj : 1u), dim3(256), 0, st, S, chunks, Bp, batch, qpart, ll, yT, G, ldg);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------------------------------------
// The same log-density in ONE launch with no scratch (round 4): a workgroup owns C chains (1 or 2), stages their dx in LDS -- plain
// vectors coalesced, tree states with the distances formed on the way -- and its 256 threads walk the FLAT stream of the matrix's
// entries, (row | column << 16, value): 12 coalesced bytes per entry, no row pointers to chase, every load independent of every other;
// an entry costs two LDS gathers per chain:   q = sum_e v_e dx[row_e] dx[col_e].
// For an exactly symmetric matrix (what `prepare`'s graphical lasso writes) the stream holds the upper triangle only, off-diagonal
// values doubled: half the bytes and half the gathers.  Thread t takes the entries t, t + 256, ... in that order, the partial sums
// are added lane by lane and wave by wave in a fixed order: bit-reproducible.  Three launches and 2 x 8 N B bytes of transposed
// scratch (above) become one launch: N = 2011 x 512 chains 34.9 -> see DESIGN.md; N = 256 x 512 chains 16.4 -> below the dense sweep.
// The value agrees with the row form above to rounding (another order of summation), not bit for bit.
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double sp_wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int C, bool TREE>
__global__ __launch_bounds__(256) void k_sparse_quad(SparseDev S, SparseTreeDev T, const double* __restrict__ X, const double* __restrict__ Rt, int64_t ld,
                                                    const double* __restrict__ tH, const double* __restrict__ rMu, int64_t batch,
                                                    double* __restrict__ ll, double* __restrict__ logjac, double* __restrict__ qout)
{
    extern __shared__ double qsh[];                          // [C][n] dx, then [C][4] the waves' partial sums
    const int n = S.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b0 = (int64_t)blockIdx.x * C;
    double* red = qsh + (size_t)C * n;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int64_t b = (b0 + c < batch) ? b0 + c : batch - 1;      // (a chain beyond the batch: the last one again, nothing stored)
        double* dx = qsh + (size_t)c * n;
        if constexpr (TREE) {
            const double* H = X + b * ld;
            const double* R = Rt + b * ld;
            const double s = tH[b] * rMu[b];
            const int rr = T.root_right;
            for (int j = tid; j < n; j += 256) {
                const int a = T.slot_node[j], pa = T.slot_parent[j];
                double d = (H[pa] - H[a]) * R[a];            // heightTreeToLengthTree, times * rates   (app/Probability.hs:201-207)
                if (j == 0) d = d + (H[0] - H[rr]) * R[rr];  // sumFirstTwo
                d = d * s;
                if (j == 0 && logjac && b0 + c < batch) logjac[b] = log(1.0 / d);          // jacobianRootBranch, :393-410
                dx[j] = d - S.mu[j];
            }
        } else {
            const double* x = X + b * ld;
            for (int j = tid; j < n; j += 256) dx[j] = x[j] - S.mu[j];
        }
    }
    __syncthreads();
    double acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0;
    const int64_t nq = S.q_nnz;
    constexpr int U = 8;                                     // entries in flight per thread
    int64_t e = tid;
    for (; e + (int64_t)(U - 1) * 256 < nq; e += (int64_t)U * 256) {
        uint32_t rc[U];
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rc[u] = S.q_rc[e + (int64_t)u * 256];
            v[u] = S.q_val[e + (int64_t)u * 256];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = (int)(rc[u] & 0xFFFFu), k = (int)(rc[u] >> 16);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = fma(v[u] * qsh[(size_t)c * n + r], qsh[(size_t)c * n + k], acc[c]);
        }
    }
    for (; e < nq; e += 256) {
        const uint32_t rc = S.q_rc[e];
        const double v = S.q_val[e];
        const int r = (int)(rc & 0xFFFFu), k = (int)(rc >> 16);
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = fma(v * qsh[(size_t)c * n + r], qsh[(size_t)c * n + k], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double w = sp_wave_sum(acc[c]);
        if (lane == 0) red[c * 4 + wave] = w;
    }
    __syncthreads();
    if (tid < C && b0 + tid < batch) {
        const double q = ((red[tid * 4 + 0] + red[tid * 4 + 1]) + red[tid * 4 + 2]) + red[tid * 4 + 3];
        if (qout) qout[b0 + tid] = q;
        if (ll) ll[b0 + tid] = S.c + (-0.5) * (S.logdet + q);       // :180 (c - 1/2 (logdet + q))
    }
}

static size_t sparse_quad_lds(int n, int C) { return sizeof(double) * ((size_t)C * n + 4 * (size_t)C); }
bool sparse_quad_available(const SparseDev& S, int64_t batch) { return S.q_rc != nullptr && S.n <= 65535 && sparse_quad_lds(S.n, 1) <= 160 * 1024 && batch >= 1; }

template <int C, bool TREE>
static hipError_t launch_quad_C(const SparseDev& S, const SparseTreeDev& T, const double* X, const double* Rt, int64_t ld, const double* tH, const double* rMu,
                                int64_t batch, double* ll, double* logjac, double* qout, hipStream_t st)
{
    const size_t lds = sparse_quad_lds(S.n, C);
    if (lds > 64 * 1024) {                                   // more than 64 KiB of LDS has to be allowed once per device
        static std::atomic<unsigned long long> allowed{0};
        int dev = 0;
        if (hipError_t e = hipGetDevice(&dev)) return e;
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!((allowed.load(std::memory_order_acquire) >> dev) & 1ull)) {
            if (hipError_t e = hipFuncSetAttribute((const void*)k_sparse_quad<C, TREE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) return e;
            allowed.fetch_or(1ull << dev, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL((k_sparse_quad<C, TREE>), dim3((unsigned)((batch + C - 1) / C)), dim3(256), lds, st, S, T, X, Rt, ld, tH, rMu, batch, ll, logjac, qout);
    return hipGetLastError();
}

hipError_t launch_sparse_quad(const SparseDev& S, const SparseTreeDev* T, const double* X, const double* Rt, int64_t ld, const double* tH,
                              const double* rMu, int64_t batch, double* ll, double* logjac, double* qout, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (!sparse_quad_available(S, batch)) return hipErrorInvalidValue;
    // two chains per workgroup share a pass over the entry stream once every CU has a workgroup anyway (and while both fit in LDS)
    const bool two = batch >= 512 && sparse_quad_lds(S.n, 2) <= 160 * 1024;
    if (T) {
        if (two) return launch_quad_C<2, true>(S, *T, X, Rt, ld, tH, rMu, batch, ll, logjac, qout, st);
        return launch_quad_C<1, true>(S, *T, X, Rt, ld, tH, rMu, batch, ll, logjac, qout, st);
    }
    if (two) return launch_quad_C<2, false>(S, SparseTreeDev{}, X, nullptr, ld, nullptr, nullptr, batch, ll, nullptr, qout, st);
    return launch_quad_C<1, false>(S, SparseTreeDev{}, X, nullptr, ld, nullptr, nullptr, batch, ll, nullptr, qout, st);
}

// which form a log-density launch takes: the one-launch quadratic form up to kSparseQuadMaxBatch chains (a workgroup per one or two chains:
// the matrix streams through every workgroup), the row form with lanes = chains beyond (the matrix once per 64 chains); mcd_set_option
// "MCD_SPARSE_QUAD" = 1 / 0 forces / forbids the former (tests, timing)
constexpr int64_t kSparseQuadMaxBatch = 4096;
static bool use_quad(const SparseDev& S, int64_t batch)
{
    const int force = opt_get(OPT_SPARSE_QUAD);
    if (!sparse_quad_available(S, batch) || force == 0) return false;
    return force == 1 || batch <= kSparseQuadMaxBatch;
}

hipError_t launch_sparse_logpdf(const SparseDev& S, const double* X, int64_t ldx, int64_t batch, double* ll, double* scratch, hipStream_t st)
{
    if (use_quad(S, batch)) return launch_sparse_quad(S, nullptr, X, nullptr, ldx, nullptr, nullptr, batch, ll, nullptr, nullptr, st);
    return launch_any<false, false>(S, SparseTreeDev{}, X, nullptr, ldx, nullptr, nullptr, batch, ll, nullptr, nullptr, 0, scratch, st);
}

hipError_t launch_sparse_grad(const SparseDev& S, const double* X, int64_t ldx, int64_t batch, double* ll, double* G, int64_t ldg, double* scratch,
                              hipStream_t st)
{
    // d ll / d x = -1/2 (P + P^T) dx: the product runs over the symmetric part (the matrix itself when it is symmetric -- what `prepare`
    // writes; for another matrix -P dx would not be the gradient of the value)
    SparseDev Sy = S;
    Sy.nnz = S.s_nnz;
    Sy.rowptr = S.s_rowptr;
    Sy.trow = S.s_trow;
    Sy.col = S.s_col;
    Sy.val = S.s_val;
    return launch_any<false, true>(Sy, SparseTreeDev{}, X, nullptr, ldx, nullptr, nullptr, batch, ll, nullptr, G, ldg, scratch, st);
}

hipError_t launch_sparse_tree_logpdf(const SparseDev& S, const SparseTreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                     const double* rMu, int64_t batch, double* ll, double* logjac, double* scratch, hipStream_t st)
{
    if (use_quad(S, batch)) return launch_sparse_quad(S, &T, H, Rt, lds, tH, rMu, batch, ll, logjac, nullptr, st);
    return launch_any<true, false>(S, T, H, Rt, lds, tH, rMu, batch, ll, logjac, nullptr, 0, scratch, st);
}

}  // namespace mcd
