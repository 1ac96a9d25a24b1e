// k_wide.hip -- log-density of many chains at once, multiply form on the fp64 matrix cores (gfx950).
//
// The column sweep of mvn_device.hpp (one chain per wave, the factor broadcast lane by lane) is the latency form: it
// wins while every chain can have a SIMD to itself.  With thousands of chains the same quantity
//     ll = c - 1/2 (logdet Sigma + |W (x - mu)|^2),  W = L^-1          (app/Probability.hs:166-173, Sigma^-1 = W^T W)
// is a triangular matrix-matrix product Z = W R with one column of R per chain, and only the column sums of Z^2 are
// wanted.  Here a workgroup owns CT tiles of 16 chains; its 8 waves own row blocks of 16 rows of W and accumulate
// Z tiles with v_mfma_f64_16x16x4_f64:
//   * W is streamed from global memory as 16 x 4 tiles stored in lane order (host_factor.cpp: pack_w_tiles), each tile
//     one coalesced 512-byte load, prefetched WD_P tiles ahead, used for CT MFMAs;
//   * R = x - mu (or the tree distances of the state minus mu) is staged once per 256-column chunk in LDS, chain-major
//     with a row stride of 258 doubles so that the 4 x 16 B-operand read is bank-conflict free;
//   * rows are taken in super blocks of 256 (16 row blocks, two per wave: slots w and 15 - w, equal work on the
//     diagonal chunk); a super block walks the chunks 0 .. s, so the accumulators never outgrow 2 CT tiles per wave and
//     N is not bounded by the register file;
//   * the result layout of the f64 MFMA keeps a chain on lane & 15 in all four result registers, so the column sums of
//     squares need no transposition: registers, two lane exchanges, then the 8 waves in a fixed order through LDS;
//   * with 16 or 32 chains per workgroup the kernel stays within 128 VGPRs and 68 KiB of LDS, so two workgroups share
//     a CU: while one stages its chunk (a cold trip to HBM) the other keeps the matrix pipe busy.
#include "wide_device.hpp"
#include <atomic>

namespace mcd {

template <int CT, bool TREE>
__global__ void __launch_bounds__(64 * WD_WAVES, CT <= 2 ? (TREE ? 3 : 4) : 2) k_wide(MvnDev M, WideSrc A, int64_t batch, double* __restrict__ ll)   // (TREE: three blocks per CU -- at four the tree prologue spilled 8 registers)
{
    extern __shared__ double smem[];
    double* rs = smem;                                   // [CT * 16][WD_LD]
    double* part = smem + CT * 16 * WD_LD;               // [WD_WAVES][CT * 16]
    double* scs = part + WD_WAVES * CT * 16;             // [CT * 16] tH * rMu per chain (tree state)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b0 = (int64_t)blockIdx.x * (CT * 16);
    const int NB = (M.n + 15) >> 4, NS = (NB + 15) >> 4;
    const int col = lane & 15, kq = lane >> 4;
    const double* __restrict__ Wt = M.Wt;

    double ssq[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) ssq[ct] = 0.0;
    WD_T(0);
    if constexpr (TREE) {
        if (tid < CT * 16) {
            const int64_t b = (b0 + tid < batch) ? b0 + tid : batch - 1;
            scs[tid] = A.tH[b] * A.rMu[b];               // :205-207  (tH * rMu)
        }
    }

    for (int s = 0; s < NS; ++s) {
        const int nb = (NB - 16 * s < 16) ? NB - 16 * s : 16;
        const int shift = 16 - nb;                        // a partial super block fills the upper slots
        const int bA = wave - shift, bB = 15 - wave - shift;
        const int64_t ibA = 16 * s + bA, ibB = 16 * s + bB;
        d4 accA[CT], accB[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) accA[ct] = accB[ct] = d4{0.0, 0.0, 0.0, 0.0};

        for (int c = 0; c <= s; ++c) {
            const int ntA = bA >= 0 ? (c < s ? WD_SB / 4 : 4 * (bA + 1)) : 0;
            const int ntB = bB >= 0 ? (c < s ? WD_SB / 4 : 4 * (bB + 1)) : 0;
            WD_T(1);
            __syncthreads();
            wide_stage<CT, TREE>(rs, scs, M, A, b0, batch, c * WD_SB, tid);
            WD_T(2);
            if constexpr (TREE) {
                if (c == 0) wide_stage_root<CT>(rs, scs, M, A, b0, batch, s == 0, tid);
            }
            __syncthreads();
            WD_T(3);
            // one row block after the other (wide_device.hpp): few registers, so that two workgroups share a CU and one
            // stages its chunk while the other keeps the matrix pipe busy
            wide_tri_pass<CT>(Wt + ((bA >= 0 ? 2 * ibA * (ibA + 1) : 0) + (WD_SB / 4) * c) * 64 + lane, ntA, 0, rs, col, kq, accA);
            wide_tri_pass<CT>(Wt + ((bB >= 0 ? 2 * ibB * (ibB + 1) : 0) + (WD_SB / 4) * c) * 64 + lane, ntB, 0, rs, col, kq, accB);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ssq[ct] = fma(accA[ct][q], accA[ct][q], ssq[ct]);
                ssq[ct] = fma(accB[ct][q], accB[ct][q], ssq[ct]);
            }
        }
    }

    WD_T(4);
    // column sums: the four lane groups of a wave, then the waves in a fixed order
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        double v = ssq[ct];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) part[wave * (CT * 16) + ct * 16 + lane] = v;
    }
    __syncthreads();
    if (tid < CT * 16 && b0 + tid < batch) {
        double q = 0.0;
#pragma unroll
        for (int w = 0; w < WD_WAVES; ++w) q += part[w * (CT * 16) + tid];
        ll[b0 + tid] = M.c + (-0.5) * (M.logdet + q);      // app/Probability.hs:169
    }
    WD_T(5);
}

// More than 64 KiB of dynamic LDS has to be allowed once per kernel and device (a process may hold handles on several
// GPUs).  mcd_mvn_create does it for every instantiation (prepare_wide), so that a first launch under stream capture needs no
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
        if (hipError_t e = hipFuncSetAttribute((const void*)k_wide<CT, TREE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)) return e;
        allowed[dev].store(true, std::memory_order_release);
    }
    return hipSuccess;
}

template <int CT, bool TREE>
static hipError_t launch_ct(const MvnDev& M, const WideSrc& A, int64_t batch, double* ll, hipStream_t st)
{
    constexpr size_t bytes = (size_t)(CT * 16 * WD_LD + WD_WAVES * CT * 16 + CT * 16) * sizeof(double);
    if (hipError_t e = allow_lds<CT, TREE>()) return e;
    const unsigned grid = (unsigned)((batch + CT * 16 - 1) / (CT * 16));
    hipLaunchKernelGGL((k_wide<CT, TREE>), dim3(grid), dim3(64 * WD_WAVES), bytes, st, M, A, batch, ll);
    return hipGetLastError();
}

template <bool TREE>
static hipError_t launch_wide(const MvnDev& M, const WideSrc& A, int64_t batch, double* ll, hipStream_t st)
{
    const int ct = wide_chain_tiles(batch);
    if (ct == 1) return launch_ct<1, TREE>(M, A, batch, ll, st);
    if (ct == 2) return launch_ct<2, TREE>(M, A, batch, ll, st);
    return launch_ct<4, TREE>(M, A, batch, ll, st);
}

hipError_t launch_logpdf_wide(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (M.Wt == nullptr) return hipErrorInvalidValue;
    WideSrc A{};
    A.X = X;
    A.ldx = ldx;
    return launch_wide<false>(M, A, batch, ll, st);
}

hipError_t launch_tree_logpdf_wide(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                   const double* rMu, int64_t batch, double* ll, double* logjac, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (M.Wt == nullptr) return hipErrorInvalidValue;
    WideSrc A{};
    A.T = T;
    A.H = H;
    A.Rt = Rt;
    A.lds = lds;
    A.tH = tH;
    A.rMu = rMu;
    A.logjac = logjac;
    return launch_wide<true>(M, A, batch, ll, st);
}

hipError_t prepare_wide()
{
    if (hipError_t e = allow_lds<1, false>()) return e;
    if (hipError_t e = allow_lds<1, true>()) return e;
    if (hipError_t e = allow_lds<2, false>()) return e;
    if (hipError_t e = allow_lds<2, true>()) return e;
    if (hipError_t e = allow_lds<4, false>()) return e;
    if (hipError_t e = allow_lds<4, true>()) return e;
    return hipSuccess;
}

}  // namespace mcd

#ifdef MCD_WIDE_STAMP
#ifndef MCD_WIDE_STAMP_GRAD
extern "C" int mcd_wide_debug_stamps(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mcd::g_wide_dbg), sizeof(unsigned long long) * mcd::WD_WAVES * 16);
}
#endif
#endif
