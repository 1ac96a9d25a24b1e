// k_split.hip -- log-density for a sampler's usual batch (1 .. 1024 chains) at 192 < N <= 256: the multiply form with the
// row blocks of W = L^-1 split over 8 workgroups per 16-chain tile (gfx950).  tools/microbench/split/README.md has the
// measurements that led here.
//
// 512 chains = 32 tiles x 8 row groups = 256 workgroups, one per CU: a workgroup takes in 1/8 of W (35 KB at N = 256)
// instead of all of it, and no dependent column chain is left.  Row group g owns the row blocks g and 15 - g (equal work),
// its 4 waves share the 68 k tiles of the two blocks, the partial z tiles are added through LDS in a fixed order.  The
// price is a reduction across workgroups: the partial sums of squares of a row group are published with relaxed
// agent-scope exchanges (performed at the coherence point; the returned value tells the thread so), a counter per tile
// is incremented after a workgroup barrier, and the row group that sees 7 adds the eight partials in a fixed order,
// writes ll and resets the counter.  No agent-scope fence: it would write the XCD's L2 back on every workgroup (measured
// 22.8 us per launch instead of 6.2).  The eight row groups of a tile share blockIdx % 8, i.e. one XCD and one L2 -- a
// latency matter, not a correctness one.  The scratch (8 x 16 partials per tile) and the counters belong to the call:
// mvn_capi.cpp keeps one set per handle and stream.
#include "wide_device.hpp"

namespace mcd {

constexpr int SP_WAVES = 4;

// N <= 256 (one chunk).  grid = 64 * ceil(tiles / 8); block = 256 threads.
__global__ void __launch_bounds__(64 * SP_WAVES) k_logpdf_split(MvnDev M, const double* __restrict__ X, int64_t ldx, int64_t batch, double* __restrict__ ll,
                                                                double* __restrict__ scratch, unsigned* __restrict__ counter)
{
    const double* __restrict__ Wt = M.Wt;
    const double* __restrict__ mu = M.mu;
    const int n = M.n;
    const double c = M.c, logdet = M.logdet;
    __shared__ double rs[16 * WD_LD];                     // residuals of the tile's 16 chains
    __shared__ double zsum[SP_WAVES][2][16][17];          // partial z tiles of the two row blocks, per wave
    __shared__ unsigned last_flag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int tile = (bid >> 6) * 8 + (bid & 7), grp = (bid >> 3) & 7;
    const int64_t b0 = (int64_t)tile * 16;
    if (b0 >= batch) return;                              // whole workgroup (tiles are dealt in groups of eight)
    const int nb = (n + 15) >> 4, shift = 16 - nb;
    const int bA = grp - shift, bB = 15 - grp - shift;
    const int col = lane & 15, kq = lane >> 4;
    // stage: 256 threads = one chain row per pass
    {
        const bool live = tid < n;
        const double m = live ? mu[tid] : 0.0;
        double v[16];
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) v[ch] = (live && b0 + ch < batch) ? X[(b0 + ch) * ldx + tid] : m;
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) rs[ch * WD_LD + tid] = v[ch] - m;
    }
    __syncthreads();
    // the k tiles of block A then block B, dealt evenly to the 4 waves
    const int ntA = bA >= 0 ? 4 * (bA + 1) : 0, ntB = bB >= 0 ? 4 * (bB + 1) : 0;
    const int total = ntA + ntB, per = ((total + SP_WAVES - 1) / SP_WAVES + 3) & ~3;   // multiples of 4
    const int lo = wave * per, hi = (lo + per < total) ? lo + per : total;
    d4 accA[1] = {d4{0.0, 0.0, 0.0, 0.0}}, accB[1] = {d4{0.0, 0.0, 0.0, 0.0}};
    if (lo < hi) {
        const int a0 = lo < ntA ? lo : ntA, a1 = hi < ntA ? hi : ntA;          // part in block A
        const int c0 = (lo > ntA ? lo : ntA) - ntA, c1 = (hi > ntA ? hi : ntA) - ntA;   // part in block B
        if (a1 > a0) wide_tri_pass<1>(Wt + ((int64_t)(2 * bA * (bA + 1)) + a0) * 64 + lane, a1 - a0, a0, rs, col, kq, accA);
        if (c1 > c0) wide_tri_pass<1>(Wt + ((int64_t)(2 * bB * (bB + 1)) + c0) * 64 + lane, c1 - c0, c0, rs, col, kq, accB);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        zsum[wave][0][kq + 4 * q][col] = accA[0][q];
        zsum[wave][1][kq + 4 * q][col] = accB[0][q];
    }
    __syncthreads();
    // z = sum over the waves (fixed order); one thread per (block, row, chain): 512 values, 256 threads x 2
    double ss = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = tid >> 4, cc = tid & 15;
        const double z = ((zsum[0][h][row][cc] + zsum[1][h][row][cc]) + zsum[2][h][row][cc]) + zsum[3][h][row][cc];
        ss = fma(z, z, ss);
    }
    // sum over the 16 rows of a chain: lanes with the same tid & 15 (stride 16 within a wave, then the 4 waves)
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    __syncthreads();
    double* part = &zsum[0][0][0][0];
    if (lane < 16) part[wave * 16 + lane] = ss;
    __syncthreads();
    if (tid < 16) {
        const double q = ((part[tid] + part[16 + tid]) + part[32 + tid]) + part[48 + tid];
        // a read-modify-write is performed in the XCD's L2 and its return tells this thread that it has been: no agent-scope
        // fence (which would write the whole L2 back: measured 22 us per launch with __threadfence()) is needed, because every
        // workgroup that touches this tile's scratch and counter runs on the same XCD
        (void)__hip_atomic_exchange(&scratch[((int64_t)tile * 8 + grp) * 16 + tid], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid == 0) last_flag = (__hip_atomic_fetch_add(&counter[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 7u) ? 1u : 0u;
    __syncthreads();
    if (last_flag && tid < 16) {
        double q = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) q += __hip_atomic_load(&scratch[((int64_t)tile * 8 + g) * 16 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b0 + tid < batch) ll[b0 + tid] = c + (-0.5) * (logdet + q);
        if (tid == 0) __hip_atomic_store(&counter[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

hipError_t launch_logpdf_split(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, double* scratch, unsigned* counter,
                               hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (M.Wt == nullptr || M.n > WD_SB || batch > kSplitMaxBatch || scratch == nullptr || counter == nullptr) return hipErrorInvalidValue;
    const int64_t tiles = (batch + 15) / 16;
    const unsigned grid = (unsigned)(((tiles + 7) / 8) * 64);
    hipLaunchKernelGGL(k_logpdf_split, dim3(grid), dim3(64 * SP_WAVES), 0, st, M, X, ldx, batch, ll, scratch, counter);
    return hipGetLastError();
}

}  // namespace mcd
