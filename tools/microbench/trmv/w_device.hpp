// w_device.hpp -- triangular MULTIPLY with the pre-inverted factor: z = W (x - mu), W = L^-1 (gfx950).
//
//   ll = c - 1/2 (logdet Sigma + || W (x - mu) ||^2),   Sigma = L L^T       (app/Probability.hs:166-173)
//
// SURVEY.md section 7 names the alternative to the dependent column sweep of a triangular SOLVE: invert the factor
// once on the host; z = W dx then is a triangular matrix-vector product without any dependency between columns (the
// reference itself multiplies with the explicit inverse Sigma^-1).  That removes the 27-cycle broadcast->FMA->
// broadcast chain per column and lets several waves share ONE chain:
//   * W is cut into row blocks of 64 rows; row block bi owns the column pairs 0 .. 32 (bi + 1) - 1.  The pairs of
//     all row blocks form one flat list that is split evenly over the NW waves of a workgroup (WPlan: at most
//     NW + R - 1 segments, each a contiguous piece of one row block and of the stream in memory).
//   * lane = row.  Per pair a wave loads 16 B per lane (1 KiB per wave instruction, PD pairs ahead in a register
//     ring), reads the two x values as an LDS BROADCAST (one ds_read_b128 with a wave-uniform address) and issues
//     two independent FMAs per chain.  No cross-lane traffic, no barrier inside the stream.
//   * Partial row sums go to LDS per segment; after one barrier each row block is summed in segment order (fixed
//     order: bit-reproducible), squared and reduced.
// All BT chains of a workgroup share every W load.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mvn_kernels.h"

namespace mcd {

typedef double wd2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double w_readlane64(double v, int l)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double w_dpp64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double w_wave_sum(double v)
{
    v += w_dpp64<0xB1>(v);
    v += w_dpp64<0x4E>(v);
    v += w_dpp64<0x124>(v);
    v += w_dpp64<0x128>(v);
    return (w_readlane64(v, 0) + w_readlane64(v, 16)) + (w_readlane64(v, 32) + w_readlane64(v, 48));
}

// Issue cursor of a wave's stream: which pair is loaded next.
struct WCursor {
    int si, s_end, pi, pi_end;
    const wd2* ptr;
};

__device__ __forceinline__ void w_cursor_seek(WCursor& cu, const double* __restrict__ stream, const WPlan& pl, int lane)
{
    if (cu.si < cu.s_end) {
        cu.pi = pl.seg_p0[cu.si];
        cu.pi_end = pl.seg_p1[cu.si];
        cu.ptr = reinterpret_cast<const wd2*>(stream) + (size_t)pl.seg_off[cu.si] * 64 + lane;
    }
}

__device__ __forceinline__ wd2 w_cursor_load(WCursor& cu, const double* __restrict__ stream, const WPlan& pl, int lane)
{
    const wd2 v = *cu.ptr;
    cu.ptr += 64;
    if (++cu.pi == cu.pi_end) {
        ++cu.si;
        w_cursor_seek(cu, stream, pl, lane);
    }
    return v;
}

// Start of a multiply pass: the first PD pairs of the wave's stream go in flight.  Called BEFORE the source vector is
// staged so that the first memory round trip of the factor overlaps the one of the chain data.
template <int PD>
__device__ __forceinline__ void w_prefetch(const double* __restrict__ stream, const WPlan& pl, int wave, int lane, WCursor& cu,
                                           wd2 (&ring)[PD])
{
    cu.si = pl.wave_seg0[wave];
    cu.s_end = pl.wave_seg0[wave + 1];
    cu.pi = cu.pi_end = 0;
    cu.ptr = nullptr;
    w_cursor_seek(cu, stream, pl, lane);
#pragma unroll
    for (int k = 0; k < PD; ++k) {
        ring[k] = wd2{0.0, 0.0};
        if (cu.si < cu.s_end) ring[k] = w_cursor_load(cu, stream, pl, lane);
    }
}

// The multiply pass of a wave over its segments.
//   xs      LDS, [BT][NPs] source vectors (NPs = 64 Rw)
//   part    LDS, [n_seg][BT][64] partial row sums
template <int BT, int PD>
__device__ __forceinline__ void w_pass(const double* __restrict__ stream, const WPlan& pl, int wave, int lane, WCursor& cu,
                                       wd2 (&ring)[PD], const double* xs, int NPs, double* part)
{
    const int s_end = pl.wave_seg0[wave + 1];
    int sc = pl.wave_seg0[wave];                       // consume cursor
    if (sc >= s_end) return;
    int pc = pl.seg_p0[sc], pc_end = pl.seg_p1[sc];
    double accA[BT], accB[BT];
#pragma unroll
    for (int c = 0; c < BT; ++c) accA[c] = accB[c] = 0.0;
    while (sc < s_end) {
#pragma unroll
        for (int k = 0; k < PD; ++k) {
            if (sc < s_end) {
                const wd2 w = ring[k];
                if (cu.si < cu.s_end) ring[k] = w_cursor_load(cu, stream, pl, lane);   // refill this slot PD pairs ahead
#pragma unroll
                for (int c = 0; c < BT; ++c) {
                    const wd2 xv = *reinterpret_cast<const wd2*>(xs + (size_t)c * NPs + 2 * pc);   // LDS broadcast
                    accA[c] = fma(w.x, xv.x, accA[c]);
                    accB[c] = fma(w.y, xv.y, accB[c]);
                }
                if (++pc == pc_end) {
#pragma unroll
                    for (int c = 0; c < BT; ++c) {
                        part[((size_t)sc * BT + c) * 64 + lane] = accA[c] + accB[c];
                        accA[c] = accB[c] = 0.0;
                    }
                    ++sc;
                    if (sc < s_end) {
                        pc = pl.seg_p0[sc];
                        pc_end = pl.seg_p1[sc];
                    }
                }
            }
        }
    }
}

// Row block bi of the product, summed over its segments in segment order (all lanes of the calling wave).
template <int BT>
__device__ __forceinline__ void w_block_rows(const WPlan& pl, int bi, int lane, const double* part, double (&z)[BT])
{
#pragma unroll
    for (int c = 0; c < BT; ++c) z[c] = 0.0;
    for (int s = pl.blk_seg0[bi]; s < pl.blk_seg0[bi + 1]; ++s)
#pragma unroll
        for (int c = 0; c < BT; ++c) z[c] += part[((size_t)s * BT + c) * 64 + lane];
}

}  // namespace mcd
