// Experiment (round 1, not wired into the library): the multiply form of the log-density with the ROW BLOCKS of W split
// over 8 workgroups per 16-chain tile, for a sampler's usual batch (512 chains = 32 tiles x 8 row groups = 256
// workgroups, one per CU).  A workgroup takes in 1/8 of W (35 KB at N = 256) instead of all of it; the price is a
// reduction across workgroups: partial sums of squares go to a scratch array, an agent-scope counter per tile tells the
// last row group to arrive that it has to add the eight partials (fixed order => reproducible bits) and write ll.
// The row groups of a tile are placed on one XCD (workgroups are dealt round-robin over the 8 XCDs: same blockIdx % 8),
// so the partials and the counter stay in one L2.
// Build / run:  see README.md in this directory.
#include "../../../mcmc-date_amd/csrc/wide_device.hpp"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>

namespace mcd {

constexpr int SP_WAVES = 4;

// N <= 256 (one chunk).  grid = tiles * 8; block = 256 threads.
__global__ void __launch_bounds__(64 * SP_WAVES) k_split(const double* __restrict__ Wt, const double* __restrict__ mu, int n, double c, double logdet,
                                                         const double* __restrict__ X, int64_t ldx, int64_t batch, double* __restrict__ ll,
                                                         double* __restrict__ scratch, unsigned* __restrict__ counter)
{
    __shared__ double rs[16 * WD_LD];                     // residuals of the tile's 16 chains
    __shared__ double zsum[SP_WAVES][2][16][17];          // partial z tiles of the two row blocks, per wave
    __shared__ unsigned last_flag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int tile = (bid >> 6) * 8 + (bid & 7), grp = (bid >> 3) & 7;
    const int64_t b0 = (int64_t)tile * 16;
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

}  // namespace mcd

// ---- stand-alone harness: random SPD problem, W tiles packed as the library does, graph of 100 launches replayed ----
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 256;
    const int B = argc > 2 ? atoi(argv[2]) : 512;
    const int reps = argc > 3 ? atoi(argv[3]) : 200;
    // L: unit-ish lower factor; W = L^-1 by forward substitution; everything in double, the check below is against the same W
    std::vector<double> L((size_t)n * n, 0.0), W((size_t)n * n, 0.0), mu(n), X((size_t)B * n);
    srand(1);
    auto rnd = []() { return rand() / (double)RAND_MAX - 0.5; };
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < i; ++j) L[(size_t)i * n + j] = 0.1 * rnd();
        L[(size_t)i * n + i] = 1.0 + 0.5 * (rnd() + 0.5);
        mu[i] = rnd();
    }
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j <= i; ++j) {
            double s = (i == j) ? 1.0 : 0.0;
            for (int k = j; k < i; ++k) s -= L[(size_t)i * n + k] * W[(size_t)k * n + j];
            W[(size_t)i * n + j] = s / L[(size_t)i * n + i];
        }
    }
    for (auto& x : X) x = rnd() * 2.0;
    const int NB = (n + 15) / 16;
    std::vector<double> Wt((size_t)2 * NB * (NB + 1) * 64, 0.0);
    for (int ib = 0; ib < NB; ++ib)
        for (int kt = 0; kt < 4 * (ib + 1); ++kt)
            for (int l = 0; l < 64; ++l) {
                const int row = 16 * ib + (l & 15), ck = 4 * kt + (l >> 4);
                if (row < n && ck <= row) Wt[((size_t)2 * ib * (ib + 1) + kt) * 64 + l] = W[(size_t)row * n + ck];
            }
    double logdet = 0.0;
    for (int i = 0; i < n; ++i) logdet += 2.0 * log(L[(size_t)i * n + i]);
    const double c = -0.9189385332046727 * n;
    const int tiles = (B + 15) / 16;
    if (tiles % 8 != 0) { fprintf(stderr, "batch must be a multiple of 128 in this experiment\n"); return 1; }
    double *dW, *dmu, *dX, *dll, *dscr;
    unsigned* dcnt;
    CK(hipMalloc(&dW, Wt.size() * 8)); CK(hipMalloc(&dmu, n * 8)); CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dll, B * 8));
    CK(hipMalloc(&dscr, (size_t)tiles * 8 * 16 * 8)); CK(hipMalloc(&dcnt, tiles * 4));
    CK(hipMemcpy(dW, Wt.data(), Wt.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dmu, mu.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(dcnt, 0, tiles * 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto launch = [&]() { hipLaunchKernelGGL(mcd::k_split, dim3(tiles * 8), dim3(256), 0, st, dW, dmu, n, c, logdet, dX, (int64_t)n, (int64_t)B, dll, dscr, dcnt); };
    launch();
    CK(hipStreamSynchronize(st));
    std::vector<double> ll(B);
    CK(hipMemcpy(ll.data(), dll, B * 8, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int b = 0; b < B; ++b) {
        double q = 0.0;
        for (int i = 0; i < n; ++i) {
            double z = 0.0;
            for (int j = 0; j <= i; ++j) z += W[(size_t)i * n + j] * (X[(size_t)b * n + j] - mu[j]);
            q += z * z;
        }
        const double ref = c - 0.5 * (logdet + q);
        worst = fmax(worst, fabs(ll[b] - ref) / fmax(1.0, fabs(ref)));
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 100; ++i) launch();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<double> ll2(B);
    CK(hipMemcpy(ll2.data(), dll, B * 8, hipMemcpyDeviceToHost));
    bool same = true;
    for (int b = 0; b < B; ++b) same = same && (ll2[b] == ll[b]);
    printf("{\"n\": %d, \"chains\": %d, \"us_per_launch\": %.3f, \"max_rel_err_vs_host_W\": %.3g, \"bits_repeat\": %s}\n", n, B, 1e3 * ms / (100.0 * reps), worst, same ? "true" : "false");
    return 0;
}
