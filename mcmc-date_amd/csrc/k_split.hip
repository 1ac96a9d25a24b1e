// k_split.hip -- log-density (raw x and tree states) for a sampler's usual batch (1 .. 1024 chains) at 128 < N <= 1024: the
// multiply form z = W (x - mu), W = L^-1, on the fp64 matrix cores with the ROW BLOCKS of W split over G workgroups per
// 16-chain tile (gfx950).  tools/microbench/split/README.md has the measurements that led here.
//
// 512 chains = 32 tiles x 8 row groups = 256 workgroups, one per CU: a workgroup takes in 1/G of W (35 KB at N = 256, 532 KB
// at N = 1024) instead of all of it, and no dependent column chain is left.  The schedule is arithmetic (split_sched.hpp):
// the row blocks are dealt to the groups in alternating direction, a group's k tiles are cut into eight equal runs (one per
// wave, two waves per SIMD) and packed once per handle in the order the waves consume them (host_factor.cpp), so a wave
// streams ONE linear run of tile pairs (16 bytes per lane and load) through a register ring whatever blocks it crosses.  A block that a
// wave holds completely is squared in registers; a block cut between waves leaves partial z tiles in LDS, added in a fixed
// order.  The residuals of the tile's 16 chains (all the columns the group needs: up to 131 KB at N = 1024) are staged once.
//
// The price is a reduction across workgroups.  Hand-over, correct under the memory model alone (no ordering between different
// locations is relied on, no placement of workgroups on XCDs -- MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement &
// inter-workgroup visibility") and free of deadlock whatever the dispatch order and residency:
//   * every partial sum has a scratch slot of its own, an 8-byte atomic variable that holds SP_POISON (a NaN pattern no
//     arithmetic produces) between launches;
//   * the first thing a row group does is to take a ticket: a RETURNING agent-scope atomic add on the tile's counter
//     (read-modify-writes on device memory are performed at the memory side, not in an XCD's L2: no add is lost whatever
//     XCDs the adders run on); the answer travels while the chain vectors are staged.  The group that draws G - 1 STARTED
//     last: it is the reader.  Every other group of the tile has started before it, is running and never waits for
//     anything -- so the reader may wait for them;
//   * a group that is not the reader stores its 16 partial sums with relaxed agent-scope atomic stores
//     (global_store_dwordx2 sc1: written through to the coherence point) and ends;
//   * the reader keeps its own partial sums in registers and reads the other G - 1 x 16 slots with relaxed agent-scope atomic
//     loads (global_load_dwordx2 sc1: never served from the CU's L1), repeating a load until it no longer sees SP_POISON -- a
//     store that is still to come or on its way is waited for, a stale value cannot be mistaken for a fresh one; then it adds
//     the partial sums in group order, writes ll, puts SP_POISON back into the slots and zero into the counter (ordered before
//     the next launch by the kernel boundary).  One memory round trip at the end of the kernel instead of the two an election
//     by the LAST ARRIVAL costs (counter add, then the loads).
// No agent-scope fence: it would write the XCD's L2 back on every workgroup (measured 22.8 us per launch instead of 6.2).
// The groups of a tile share blockIdx % 8, i.e. in practice one XCD and one L2: a matter of speed only (the chain vectors
// are fetched once per XCD); MCD_SPLIT_SCATTER=1 deals a tile's groups to consecutive workgroups -- different XCDs -- and the
// stress test runs both.  The scratch (G x 16 slots per tile, a counter per tile) belongs to the stream the call is made
// on (or to the capture while a stream is being captured): SplitHost keeps the sets.
#include "wide_device.hpp"
#include "host_factor.h"
#include "split_sched.hpp"
#include "options.h"

#include <stdlib.h>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <functional>
#include <utility>
#include <vector>

namespace mcd {

static_assert(SP_MAXSEG == SPH_MAXSEG, "host_factor.h and mvn_kernels.h differ");

constexpr unsigned long long SP_POISON = 0x7FF8C0DEDEADBEEFull;   // "no partial sum yet": a quiet NaN with a payload no arithmetic produces

constexpr int SP_Z = 256;                                  // doubles of one partial z tile (16 rows x 16 chains)
template <int NC>
constexpr size_t split_lds_bytes()
{
    // NC = 4 fills the CU's 160 KiB: there the per-wave sums go where the residuals were (an extra barrier), not beside them
    return (size_t)(16 * (NC * 256 + 2) + SP_NSLOT * SP_Z + (NC < 4 ? SP_NW * 64 : 0) + 8) * sizeof(double);
}

typedef double d2 __attribute__((ext_vector_type(2)));

// diagnostic build (make stamp_split): s_memtime phase stamps of every wave of workgroup 0 and of the workgroup that came
// last for tile 0, read back with mcd_split_debug_stamps (tools/microbench/split_stamps.py)
#ifdef MCD_SPLIT_STAMP
__device__ unsigned long long g_split_dbg[2 * SP_NW * 16];
// stamps stay in scalar registers until the wave ends (a store per stamp would put waits into the phases being timed)
#define SP_T(i) do { spt[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define SP_T_FLUSH() do { if (tile == 0 && (grp == 0 || grp == G - 1) && lane == 0) { \
        for (int i_ = 0; i_ < 16; ++i_) g_split_dbg[((grp == 0 ? 0 : 1) * SP_NW + wave) * 16 + i_] = spt[i_]; } } while (0)
#else
#define SP_T(i) do { } while (0)
#define SP_T_FLUSH() do { } while (0)
#endif
// tile PAIRS in flight per wave.  N <= 256: 8 (a wave's whole run: at most 4 groups of 4 tiles).  Far above, a wave streams hundreds
// of tiles and what bounds it is the data in flight: a CU takes in 33-41 B/clk from L2 / Infinity Cache at
// a latency of some 3 000 cycles, i.e. ~115 KB have to be under way per CU to keep that rate; 8 waves x 8 pairs are 64 KB,
// 16 pairs 128 KB.  Measured (tools/gpu/ring2.sh, HIP events, 512 chains): N = 1024 21.3 -> 20.0 us, its gradient 42.2 -> 40.4;
// N = 384 ... 768 and the tree states level or 1-4 % slower with 16 (more registers, a longer prologue): 16 only at NC = 4.
template <int NC>
constexpr int split_ring()
{
    return NC == 4 ? 16 : 8;
}

// One group: 4 tiles = the ring slots 2 PH and 2 PH + 1 against the four B operands in `bc`.  Software pipeline, written out
// because the compiler's own ordering (every load as early as possible) makes a refilled slot overlap the value still waiting
// for its MFMA, which costs a register copy and a full vmcnt drain at every loop end: the group first refills the two slots
// the PREVIOUS group consumed (dead by now: the refill lands in the same registers) with the pairs RING - 2 and RING - 1 ahead,
// reads the NEXT group's B operands from LDS into `bn`, then issues its own four MFMAs, whose operands were requested
// RING / 2 - 1 groups (A) and one group (B) ago; nothing moves across the group boundary.
template <int RING, int PH>
__device__ __forceinline__ void split_group(const d2* __restrict__ w, d2 (&ring)[RING], int jp, int lastp, const double* rk_next,
                                            const double (&bc)[4], double (&bn)[4], d4& acc)
{
    constexpr int NP = RING / 2, PR = (PH + NP - 1) % NP;
    if (jp + RING - 1 <= lastp) {                          // wave-uniform: no request past the end of the run (the loop end waits
#pragma unroll                                             // for everything in flight: a useless load would cost a round trip)
        for (int p = 0; p < 2; ++p) ring[2 * PR + p] = w[(jp + RING - 2 + p) * 64];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) bn[i] = rk_next[i * 4];
    // one accumulator: a SIMD completes one v_mfma_f64_16x16x4_f64 per 64 cycles (26.9 ns: 77.9 TFLOP/s over the chip, the fp64
    // matrix peak) whether the instructions depend on each other or not and whether one wave or two issue them
    // (tools/microbench/mfma64, wall clock)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[2 * PH].x, bc[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[2 * PH].y, bc[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[2 * PH + 1].x, bc[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[2 * PH + 1].y, bc[3], acc, 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);     // 2 VMEM reads, then
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     // 2 DS reads (ds_read2_b64), then
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);     // 4 MFMAs
}

// the groups K .. COUNT - 1 of one trip round the ring (phases PH + K, compile time), B operands alternating between b0 and b1
template <int RING, int PH, int K, int COUNT>
__device__ __forceinline__ void split_round(const d2* __restrict__ w, d2 (&ring)[RING], int jp, int lastp, const double* rk, double (&b0)[4],
                                            double (&b1)[4], d4& acc)
{
    if constexpr (K < COUNT) {
        constexpr int NP = RING / 2;
        if constexpr (K % 2 == 0)
            split_group<RING, (PH + K) % NP>(w, ring, jp + 2 * K, lastp, rk + 16 * (K + 1), b0, b1, acc);
        else
            split_group<RING, (PH + K) % NP>(w, ring, jp + 2 * K, lastp, rk + 16 * (K + 1), b1, b0, acc);
        split_round<RING, PH, K + 1, COUNT>(w, ring, jp, lastp, rk, b0, b1, acc);
    }
}
// the last `left` < RING / 2 groups of a run
template <int RING, int PH, int K>
__device__ __forceinline__ void split_tail(const d2* __restrict__ w, d2 (&ring)[RING], int jp, int lastp, int left, const double* rk,
                                           double (&b0)[4], double (&b1)[4], d4& acc)
{
    constexpr int NP = RING / 2;
    if constexpr (K < NP - 1) {
        if (K < left) {
            if constexpr (K % 2 == 0)
                split_group<RING, (PH + K) % NP>(w, ring, jp + 2 * K, lastp, rk + 16 * (K + 1), b0, b1, acc);
            else
                split_group<RING, (PH + K) % NP>(w, ring, jp + 2 * K, lastp, rk + 16 * (K + 1), b1, b0, acc);
            split_tail<RING, PH, K + 1>(w, ring, jp, lastp, left, rk, b0, b1, acc);
        }
    }
}

// ng groups starting at phase PH (compile time); jp = pair index of the first group; rk = this lane's B operand of the first
// k tile.  The B operands one group past the end of the run are read and dropped (LDS behind rs is the kernel's own).
template <int RING, int PH>
__device__ __forceinline__ void split_run(const d2* __restrict__ w, d2 (&ring)[RING], int jp, int lastp, int ng, const double* rk, d4& acc)
{
    constexpr int NP = RING / 2;
    double b0[4], b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b0[i] = rk[i * 4];
    __builtin_amdgcn_sched_barrier(0);
    int g = 0;
    for (; g + NP <= ng; g += NP) {
        split_round<RING, PH, 0, NP>(w, ring, jp, lastp, rk, b0, b1, acc);
        jp += 2 * NP;
        rk += 16 * NP;
    }
    if (g < ng) split_tail<RING, PH, 0>(w, ring, jp, lastp, ng - g, rk, b0, b1, acc);
}

// grid = ceil(tiles / 8) * 8 * G; block = 576 threads: 8 working waves (two per SIMD) and a helper wave, which takes the
// group's ticket and works out which row blocks are cut between waves -- a memory round trip and a few hundred scalar
// instructions that would otherwise sit on the working waves' critical path -- and leaves both in LDS.
// The helper wave exists where a kernel is a few microseconds long (NC = 1: N <= 256); above, where a ninth wave would cost the
// other eight a third of their registers, wave 0 does both jobs behind its staging loads.
template <int NC>
constexpr int split_threads()
{
    return 64 * (SP_NW + (NC == 1 ? 1 : 0));
}
// MODE 0: ln density.  MODE 1: ln density, and the z tiles are written to G_.zt (first half of the gradient).  MODE 2: second
// half of the gradient: the chain vectors are the z rows of G_.zt in REVERSED order and S is the schedule of J W^T J (J = the
// reversal: lower triangular like W), so the same walk computes J y = (J W^T J)(J z), y = W^T z = Sigma^-1 (x - mu); the y
// tiles leave as d ll / d x = -y in the caller's rows (G_.t == 0) or as they are in the tile-major scratch (G_.t == 1: tree
// states, k_split_tree_chain applies the chain rule).  No sum, no hand-over in MODE 2.
struct SplitGrad {
    double* zt;        // [tile][16 NB][16 chains] z rows, tile-major (a z tile = 256 consecutive doubles: coalesced both ways)
    double* out;       // MODE 2: G [batch][ldo] (t == 0) or the y scratch, laid out like zt (t == 1)
    int64_t ldo;
    int t;
};

template <int NC, bool TREE, int MODE>
__global__ void __launch_bounds__(split_threads<NC>(), NC == 1 ? 3 : 2) k_split(MvnDev M, SplitSched S, WideSrc A, SplitGrad G_, int64_t batch, int flags,
                                                                        double* __restrict__ ll, double* scratch, unsigned* counter)
{
    constexpr int LD = NC * 256 + 2;                       // LDS row stride in doubles: = 4 dwords (mod 64 banks)
    extern __shared__ double smem[];
    double* rs = smem;                                     // [16][LD] residuals of the tile's chains
    double* zsum = rs + 16 * LD;                           // [SP_NSLOT][SP_Z] partial z tiles of the blocks cut between waves
    double* part = NC < 4 ? zsum + SP_NSLOT * SP_Z : rs;   // [SP_NW][64] (NC = 4: over the residuals, once every wave is done with them)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = S.G;
    int tile, grp;
    {
        const int bid = blockIdx.x;
        if (flags & 1) {                                   // test layout: a tile's groups on consecutive workgroups (different XCDs)
            tile = bid / G;
            grp = bid - tile * G;
        } else {                                           // a tile's groups share blockIdx % 8
            const int sup = bid / (8 * G), rem = bid - sup * 8 * G;
            tile = sup * 8 + (rem & 7);
            grp = rem >> 3;
        }
    }
    const int64_t b0 = (int64_t)tile * 16;
    if (b0 >= batch) return;                               // whole workgroup
#ifdef MCD_SPLIT_STAMP
    unsigned long long spt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    int* aux = reinterpret_cast<int*>((NC < 4 ? zsum + SP_NSLOT * SP_Z + SP_NW * 64 : zsum + SP_NSLOT * SP_Z));   // [16]: ticket, cuts
    const int rot = (flags & 2) ? 0 : 2 * grp;             // first chain a group stages (experiment knob: bit 1 = no rotation)
    const int NB = S.NB;
    const SpGroup sg = sp_group(NB, G, grp);               // split_sched.hpp: scalar arithmetic on blockIdx, no table to fetch
    const bool HELPER = NC == 1 && blockDim.x > 64 * SP_NW;   // (the launcher leaves the helper wave out when two workgroups have to share a CU)
    // Who will add up the tile's partial sums: the row group that STARTS last (see the header) -- a returning atomic add,
    // whose answer travels while the chain vectors are staged; and which row blocks are cut between waves.  Both go to LDS.
    auto ticket_and_cuts = [&]() {
        unsigned ticket = 0;
        if constexpr (MODE != 2) {
            if (lane == 0) ticket = __hip_atomic_fetch_add(&counter[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {                                   // the cut blocks come with the schedule (kernel argument)
            aux[0] = (int)ticket;
#pragma unroll
            for (int x = 0; x < SP_NW; ++x) aux[1 + x] = S.cuts[grp][x];
        }
    };
    if (HELPER && wave == SP_NW) {                         // the helper wave
        if (flags & 4) return;
        ticket_and_cuts();
        __syncthreads();                                   // staging done (the working waves' first barrier)
        if (flags & 8) return;
        __syncthreads();                                   // runs done
        return;
    }
    const int lo = sp_run_start(sg, wave);
    const int T = sp_run_start(sg, wave + 1) - lo;
    const int ncols = sg.ncols;
    const int col = lane & 15, kq = lane >> 4;
    if (flags & 4) return;                                 // timing probe (MCD_SPLIT_PROBE): the launch alone
    SP_T(0);

    // the first tiles of this wave's run are requested before anything else: they travel while the chain vectors do.  Tiles
    // are stored in pairs (lane l: its element of tile 2 i, then of tile 2 i + 1), so one 16-byte load per lane brings two
    const int npair = T >> 1, lastp = npair > 0 ? npair - 1 : 0;
    const d2* __restrict__ w = reinterpret_cast<const d2*>(S.Ws) + (int64_t)((S.base[grp] + lo) >> 1) * 64 + lane;
    // (N <= 256, where a microsecond counts.  Above, the staging needs the registers -- a ring kept alive across it gets
    // spilled, and reloaded inside the multiply loop -- and the first tiles are requested once the staging loads are done.)
    constexpr int RING = split_ring<NC>();
    d2 ring[RING];
    if constexpr (NC == 1) {
#pragma unroll
        for (int p = 0; p < RING; ++p) ring[p] = w[(p < lastp ? p : lastp) * 64];
    }

    // a run that fits the ring (at most 4 groups of 4 tiles: N <= 352 with 8 row groups) is multiplied by straight-line code
    // with the ring slots as they stand; what each group needs -- its first k tile, and whether it ends a segment and how --
    // is worked out here, under the latency of the loads above
    const bool fits = NC == 1 && npair <= RING;            // (above N = 256 the general walk serves short runs as well)
    int gk[4] = {0, 0, 0, 0}, ge[4] = {-1, -1, -1, -1};
    if (fits && T > 0) {
        int gi = 0;
        sp_for_each_segment(NB, G, grp, sg, wave, [&](int k0, int nt, int kind) {
            const int ng = nt >> 2;
            for (int j = 0; j < ng; ++j, ++gi) {
                const int k = k0 + 4 * j, e = (j == ng - 1) ? kind : -1;
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    if (gi == x) {
                        gk[x] = k;
                        ge[x] = e;
                    }
            }
        });
    }

    // ---- stage the residuals
    const int NR = 16 * NB;                                // rows of a tile in the tile-major scratch
    if constexpr (MODE == 2) {
        // the z rows of this tile, reversed: rs[ch][k] = z[n - 1 - k][ch]; thread = (row, chain), chains fastest (128-byte rows).
        // Chains beyond the batch hold exact zeros already (their residuals were zeros), rows beyond N are forced to zero.
        const double* __restrict__ zt = G_.zt + (int64_t)tile * NR * 16;
        const int ch = tid & 15, k0 = (tid & 511) >> 4;    // (the helper wave never comes here)
        const int nlast = M.n - 1;
#pragma unroll 1
        for (int it = 0; it < NC; ++it) {
            double v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = (it * 8 + i) * 32 + k0;
                const int kc = k < nlast ? k : nlast;
                v[i] = zt[(nlast - kc) * 16 + ch];
            }
            if (!HELPER && wave == 0 && it == 0) ticket_and_cuts();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = (it * 8 + i) * 32 + k0;
                if (k < ncols) rs[ch * LD + k] = (k < M.n) ? v[i] : 0.0;
            }
        }
    } else if constexpr (!TREE) {
        // thread = column ((tid & 255) + 256 c), every second chain.  Every load is unconditional on a clamped address (a load
        // behind a condition costs a branch and its own wait each); padding is then forced to exact zeros by the select.  The
        // groups of a tile start on different chains, so that they do not ask one L2 channel for the same line at the same time.
        double m[NC], v[NC][8];
        const int nlast = M.n - 1, tc = tid & 255, ch0 = tid >> 8;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = c * 256 + tc;
            const int kc = k < nlast ? k : nlast;
            m[c] = M.mu[kc];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ch = (2 * i + ch0 + rot) & 15;
                const int64_t b = (b0 + ch < batch) ? b0 + ch : batch - 1;
                const double* __restrict__ rowp = A.X + b * A.ldx;              // wave-uniform row base, per-lane 32-bit column
                v[c][i] = rowp[kc];
            }
        }
        if (!HELPER && wave == 0) ticket_and_cuts();       // behind the loads: its wait coincides with theirs
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = c * 256 + tc;
            if (k < ncols) {
                const bool live = k < M.n;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int ch = (2 * i + ch0 + rot) & 15;
                    const double d = v[c][i] - m[c];                            // dxs = xs - mu  (app/Probability.hs:171)
                    rs[ch * LD + k] = (live && b0 + ch < batch) ? d : 0.0;      // padded columns and chains: exact zeros
                }
            }
        }
    } else {
        // distances from the tree state -- app/Probability.hs:201-207, the arithmetic of load_tree (mvn_device.hpp) and
        // wide_stage (wide_device.hpp): d = ((h_parent - h_node) * rate) * (tH * rMu), slot 0 = the two root branches added
        // (sumFirstTwo, app/Tools.hs:36-48).  A wave owns two chains: a chain's heights arrive as one contiguous row
        // (coalesced) in the chain's own LDS row, the parent / node heights of a slot are gathered from there -- a gather from
        // global memory costs a 128-byte line per lane -- and the distances then overwrite the row (one wave, LDS operations
        // in order: no barrier).  The rates are read by slot: slot k's node is k + 1 or k + 2, nearly contiguous.
        constexpr int NI = NC * 4;                         // columns per lane: k = 64 i + lane
        const int nn = A.T.n_nodes;
        int na[NI], npa[NI];
        double m[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int k = i * 64 + lane;
            const int kc = k < M.n - 1 ? k : M.n - 1;     // unconditional loads on clamped addresses; `live` decides later
            na[i] = A.T.slot_node[kc];
            npa[i] = A.T.slot_parent[kc];
            m[i] = M.mu[kc];
        }
        const int rr = A.T.root_right;
        constexpr int CB = NC >= 4 ? 1 : 2;                // chains whose loads are in flight together per wave (registers)
#pragma unroll 1
        for (int j0 = 0; j0 < 2; j0 += CB) {
            double hv[CB][NI + 1], ra[CB][NI], sc[CB], rrr[CB];
#pragma unroll
            for (int jj = 0; jj < CB; ++jj) {
                const int ch = (8 * (j0 + jj) + wave + rot) & 15;      // the groups of a tile start on different chains
                const int64_t b = (b0 + ch < batch) ? b0 + ch : batch - 1;
                const double* __restrict__ h = A.H + b * A.lds;
                const double* __restrict__ r = A.Rt + b * A.lds;
#pragma unroll
                for (int i = 0; i <= NI; ++i) {
                    const int v = i * 64 + lane;
                    hv[jj][i] = h[v < nn ? v : nn - 1];
                }
#pragma unroll
                for (int i = 0; i < NI; ++i) ra[jj][i] = r[na[i]];
                rrr[jj] = r[rr];
                sc[jj] = A.tH[b] * A.rMu[b];
            }
            if (!HELPER && wave == 0 && j0 == 0) ticket_and_cuts();   // behind the loads: its wait coincides with theirs
#pragma unroll
            for (int jj = 0; jj < CB; ++jj) {
                const int ch = (8 * (j0 + jj) + wave + rot) & 15;
                const bool in = b0 + ch < batch;
                double* row = rs + ch * LD;
#pragma unroll
                for (int i = 0; i <= NI; ++i) {
                    const int v = i * 64 + lane;
                    if (v < LD) row[v] = hv[jj][i];
                }
                double ha[NI], hp[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    ha[i] = row[na[i]];
                    hp[i] = row[npa[i]];
                }
                const double h0 = row[0], hr = row[rr];
                __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): every height has been read before the row is overwritten
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int k = i * 64 + lane;
                    double d = (hp[i] - ha[i]) * ra[jj][i];
                    if (k == 0) {                          // the root slot: both root branches; the root-branch Jacobian
                        d = d + (h0 - hr) * rrr[jj];
                        d = d * sc[jj];
                        if (grp == 0 && in && A.logjac != nullptr) A.logjac[b0 + ch] = log(1.0 / d);   // app/Probability.hs:394, 409
                    } else {
                        d = d * sc[jj];
                    }
                    if (k < ncols) row[k] = (in && k < M.n) ? d - m[i] : 0.0;     // padded columns and chains: exact zeros
                }
            }
        }
    }
    if constexpr (NC > 1) {
#pragma unroll
        for (int p = 0; p < RING; ++p) ring[p] = w[(p < lastp ? p : lastp) * 64];
    }
    SP_T(1);
    __syncthreads();
    if (flags & 8) return;                                 // timing probe: launch + staging

    // ---- this wave's run of tiles: z tile += W tile x R tile, segment after segment.  A segment (a run of k tiles inside
    // one row block) is a whole number of groups of 4 tiles = 2 ring slots; the ring has RING slots, so a group's slots
    // are known at compile time once its phase (group index mod RING / 2) is: split_run is instantiated per starting phase
    double ss = 0.0;                                       // sum of squares of the blocks this wave holds completely
    SP_T(2);
    // a finished z tile (block blk) leaves the workgroup (MODE 1, 2): lane = (chain col, row quarter kq), rows kq + 4 q
    auto emit = [&](const d4& z, int blk) {
        if constexpr (MODE == 1) {
            double* p = G_.zt + ((int64_t)tile * NR + 16 * blk) * 16 + lane;       // (kq + 4 q) * 16 + col = lane + 64 q
#pragma unroll
            for (int q = 0; q < 4; ++q) p[64 * q] = z[q];
        } else if constexpr (MODE == 2) {
            if (G_.t) {
                double* p = G_.out + ((int64_t)tile * NR + 16 * blk) * 16 + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) p[64 * q] = z[q];
            } else if (b0 + col < batch) {
                double* p = G_.out + (b0 + col) * G_.ldo;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int jr = 16 * blk + kq + 4 * q;                          // row of J y
                    if (jr < M.n) p[M.n - 1 - jr] = -z[q];                         // d ll / d x = -Sigma^-1 (x - mu)
                }
            }
        }
    };
    auto flush = [&](d4& acc, int kind, int blk) {         // a segment ends: square a whole block, park a partial z tile
        if (kind == 0) {
            if constexpr (MODE != 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) ss = fma(acc[q], acc[q], ss);
            }
            emit(acc, blk);
        } else {
            double* zs = zsum + (kind == 1 ? 2 * wave - 1 : 2 * wave) * SP_Z;
#pragma unroll
            for (int q = 0; q < 4; ++q) zs[(kq + 4 * q) * 16 + col] = acc[q];            // f64 result layout: row = (lane >> 4) + 4 reg
        }
        acc = d4{0.0, 0.0, 0.0, 0.0};
    };
    if (fits) {
        if (T > 0) {
            const double* rb = rs + col * LD + kq;
            const int ngr = T >> 2;
            double b[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int i = 0; i < 4; ++i) b[g][i] = rb[(gk[g] + i) * 4];              // (groups past the end read k tile 0 and drop it)
            }
            d4 acc = d4{0.0, 0.0, 0.0, 0.0};
            SP_T(8);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g == 1) SP_T(9);
                if (g == 2) SP_T(10);
                if (g == 3) SP_T(11);
                if (g < ngr) {
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[2 * g].x, b[g][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[2 * g].y, b[g][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[2 * g + 1].x, b[g][2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[2 * g + 1].y, b[g][3], acc, 0, 0, 0);
                    if (ge[g] >= 0) flush(acc, ge[g], gk[g] >> 2);          // (a whole block's last group starts at k = 4 blk)
                }
            }
            SP_T(12);
        }
    } else if (T > 0) {
        const double* rb = rs + col * LD + kq;
        int jp = 0, ph = 0;                                // pair index of the next tile pair to be used, its phase
        sp_for_each_segment(NB, G, grp, sg, wave, [&](int k0, int nt, int kind) {
            const int ng = nt >> 2;
            d4 acc = d4{0.0, 0.0, 0.0, 0.0};
            const double* rk = rb + k0 * 4;
            if constexpr (RING == 8) {
                switch (ph) {
                case 0: split_run<RING, 0>(w, ring, jp, lastp, ng, rk, acc); break;
                case 1: split_run<RING, 1>(w, ring, jp, lastp, ng, rk, acc); break;
                case 2: split_run<RING, 2>(w, ring, jp, lastp, ng, rk, acc); break;
                default: split_run<RING, 3>(w, ring, jp, lastp, ng, rk, acc); break;
                }
            } else {
                switch (ph) {
                case 0: split_run<RING, 0>(w, ring, jp, lastp, ng, rk, acc); break;
                case 1: split_run<RING, 1>(w, ring, jp, lastp, ng, rk, acc); break;
                case 2: split_run<RING, 2>(w, ring, jp, lastp, ng, rk, acc); break;
                case 3: split_run<RING, 3>(w, ring, jp, lastp, ng, rk, acc); break;
                case 4: split_run<RING, 4 % (RING / 2)>(w, ring, jp, lastp, ng, rk, acc); break;
                case 5: split_run<RING, 5 % (RING / 2)>(w, ring, jp, lastp, ng, rk, acc); break;
                case 6: split_run<RING, 6 % (RING / 2)>(w, ring, jp, lastp, ng, rk, acc); break;
                default: split_run<RING, 7 % (RING / 2)>(w, ring, jp, lastp, ng, rk, acc); break;
                }
            }
            jp += 2 * ng;
            ph = (ph + ng) & (RING / 2 - 1);
            flush(acc, kind, (nt >> 2) - 1);               // (kind 0: the segment is the whole block, 4 (blk + 1) tiles)
        });
    }
    SP_T(3);
    if constexpr (MODE != 2) {
        if constexpr (NC >= 4) __syncthreads();
        part[wave * 64 + lane] = ss;                       // the blocks this wave squared itself: lane = (chain, row quarter)
    }
    __syncthreads();

    // ---- everything else happens in wave 0 (one barrier instead of three): lane = (chain col, row quarter kq), the layout of
    // the MFMA result.  First the whole blocks of the eight waves, then the blocks cut between waves: the partial z tiles are
    // added in wave order, squared, and added over the lane's four rows; all in a fixed order.
    if (wave != 0) {
        SP_T_FLUSH();
        return;
    }
    SP_T(13);
    double tot = 0.0;
    if constexpr (MODE != 2) {
        double pv[SP_NW];
#pragma unroll
        for (int x = 0; x < SP_NW; ++x) pv[x] = part[x * 64 + lane];
#pragma unroll
        for (int x = 0; x < SP_NW; ++x) tot += pv[x];
    }
    const int ncut = aux[1];
#pragma unroll 1
    for (int ci = 0; ci < ncut; ++ci) {
        const int cw = aux[2 + ci], wf = cw & 15, wl = (cw >> 4) & 15;
        d4 z;
        const double* z0 = zsum + 2 * wf * SP_Z + kq * 16 + col;
#pragma unroll
        for (int q = 0; q < 4; ++q) z[q] = z0[64 * q];
        for (int x = wf + 1; x <= wl; ++x) {               // partial z tiles in wave order
            const double* zx = zsum + (2 * x - 1) * SP_Z + kq * 16 + col;
            double t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = zx[64 * q];
#pragma unroll
            for (int q = 0; q < 4; ++q) z[q] += t[q];
        }
        if constexpr (MODE != 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tot = fma(z[q], z[q], tot);
        }
        emit(z, cw >> 8);
    }
    SP_T(14);
    if constexpr (MODE == 2) {
        SP_T_FLUSH();
        return;
    }
    // the four row quarters of a chain sit on the lanes col, col + 16, col + 32, col + 48
    tot += __shfl_xor(tot, 16);
    tot += __shfl_xor(tot, 32);
    SP_T(4);
    if (flags & 16) return;                                // timing probe: everything but the hand-over
    // ---- hand-over (see the header): all of it inside wave 0
    {
        unsigned long long* slots = reinterpret_cast<unsigned long long*>(scratch) + (int64_t)tile * G * 16;
        const double q = tot;
        const unsigned ticket = (unsigned)__builtin_amdgcn_readfirstlane(aux[0]);   // taken by the helper wave at the start
        const bool reader = ticket == (unsigned)(G - 1);   // every other group of the tile had started before this one
        if (!reader) {
            if (lane < 16) {
                unsigned long long bits = (unsigned long long)__double_as_longlong(q);
                if (bits == SP_POISON) bits = 0x7FF8000000000000ull;             // (a NaN input with this very payload: any NaN will do)
                __hip_atomic_store(&slots[grp * 16 + lane], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            SP_T(5);
        } else {
            SP_T(5);
            if (lane < 16) {
                // the G - 1 other groups are running (they took their tickets before this one) and wait for nothing: each of
                // their slots will leave SP_POISON.  All loads in flight together, 8 at a time, in group order (fixed order of
                // the sum); a slot still empty is polled again.
                double sum = 0.0;
                bool lost = false;
                for (int g0 = 0; g0 < G; g0 += 8) {
                    unsigned long long bits[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        bits[i] = __hip_atomic_load(&slots[(g0 + i) * 16 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (g0 + i == grp) {
                            sum += q;                                           // this group's own partial sum never left the registers
                            continue;
                        }
                        for (int spin = 0; bits[i] == SP_POISON && spin < (1 << 24); ++spin) {   // bounded: never hang
                            __builtin_amdgcn_s_sleep(2);
                            bits[i] = __hip_atomic_load(&slots[(g0 + i) * 16 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        lost |= bits[i] == SP_POISON;
                        sum += __longlong_as_double((long long)bits[i]);
                        __hip_atomic_store(&slots[(g0 + i) * 16 + lane], SP_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (lost) sum = __longlong_as_double(0x7FF8000000000000ll);
                if (b0 + lane < batch) ll[b0 + lane] = M.c + (-0.5) * (M.logdet + sum);   // app/Probability.hs:169
            }
            if (lane == 0) __hip_atomic_store(&counter[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        SP_T(6);
        SP_T_FLUSH();
    }
}

// Chain rule from g = d ll / d distances (by slot, = -y) to the tree state (as k_tree_grad.hip):
// d ll / d r_v = s g t_v, d ll / d h_v = s (sum over children c of g_c r_c - g_v r_v), d ll / d tH = g.d / tH, d ll / d rMu = g.d / rMu.
// One workgroup per chain; y comes from the tile-major scratch MODE 2 wrote (rows of J y: slot j is row n - 1 - j);
// e_v = s g r_v is exchanged through LDS by node id.
__global__ void __launch_bounds__(1024) k_split_tree_chain(int n, int NR, TreeDev T, const double* __restrict__ H, const double* __restrict__ Rt,
                                                           int64_t lds, const double* __restrict__ tH, const double* __restrict__ rMu,
                                                           const double* __restrict__ yt, double* __restrict__ gH, double* __restrict__ gR,
                                                           double* __restrict__ gtH, double* __restrict__ grMu)
{
    // (as many threads as slots where possible -- the launcher picks 256, 512 or 1024: the kernel is a chain of dependent
    // loads, slot -> node -> height, node -> children -> e, and one trip through it costs the same for 1 or 4 slots per thread)
    extern __shared__ double smem[];
    double* e = smem;                                      // [n_nodes]
    double* red = smem + T.n_nodes_pad;                    // [16]
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nt = blockDim.x;
    const double* __restrict__ y = yt + ((b >> 4) * NR) * 16 + (b & 15);
    const double* __restrict__ h = H + b * lds;
    const double* __restrict__ r = Rt + b * lds;
    const double s = tH[b] * rMu[b];
    // this thread's first node: its children are looked up now, under the latency of the slot loads
    const int v0 = tid < T.n_nodes ? tid : T.n_nodes - 1;
    const int p0 = T.child_ptr[v0], p1 = T.child_ptr[v0 + 1];
    const int c0 = p1 > p0 ? T.child_idx[p0] : 0, c1 = p1 > p0 + 1 ? T.child_idx[p0 + 1] : 0;
    double gd = 0.0;
    for (int j = tid; j < n; j += nt) {
        const int a = T.slot_node[j], pa = T.slot_parent[j];
        const double g = -y[(n - 1 - j) * 16];
        const double t = h[pa] - h[a], ra = r[a], sg = s * g;
        gR[b * lds + a] = sg * t;
        e[a] = sg * ra;
        gd += g * ((t * ra) * s);
    }
    if (tid == 0) {                                        // the second root branch shares slot 0; the root has no branch
        const int rr = T.root_right;
        const double r2 = r[rr], t2 = h[0] - h[rr], g0 = -y[(n - 1) * 16], sg = s * g0;
        gR[b * lds + rr] = sg * t2;
        gR[b * lds] = 0.0;                                 // stem rate: unused by the likelihood
        e[rr] = sg * r2;
        e[0] = 0.0;
        gd += g0 * ((t2 * r2) * s);
    }
    gd = wd_wave_sum(gd);
    if (lane == 0) red[wave] = gd;
    __syncthreads();
    if (tid == 0) {
        double gdot = 0.0;
        for (int w = 0; w < (nt >> 6); ++w) gdot += red[w];
        gtH[b] = gdot / tH[b];
        grMu[b] = gdot / rMu[b];
    }
    if (tid < T.n_nodes) {
        double acc = (tid == 0) ? 0.0 : -e[tid];
        if (p1 > p0) acc += e[c0];
        if (p1 > p0 + 1) acc += e[c1];
        for (int ci = p0 + 2; ci < p1; ++ci) acc += e[T.child_idx[ci]];
        gH[b * lds + tid] = acc;
    }
    for (int v = tid + nt; v < T.n_nodes; v += nt) {
        double acc = -e[v];
        for (int ci = T.child_ptr[v]; ci < T.child_ptr[v + 1]; ++ci) acc += e[T.child_idx[ci]];
        gH[b * lds + v] = acc;
    }
}

__global__ void k_split_poison(unsigned long long* slots, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slots[i] = SP_POISON;
}

// ------------------------------------------------------------------------------------------------------------------
// host side: schedules per handle, scratch sets per stream
// ------------------------------------------------------------------------------------------------------------------
struct SplitHost {
    int n = 0, device = 0;
    SplitSched sched[3] = {};                              // G = 8, 16, 32 (G = 0: not built for this N)
    SplitSched sched_b[3] = {};                            // the same for J W^T J (second half of the gradient)
    size_t zt_doubles = 0;                                 // kSplitMaxBatch chains x 16 NB rows
    std::vector<void*> dev_allocs;
    struct Set {
        double* partials = nullptr;
        unsigned* counters = nullptr;
        double* zt = nullptr;                              // gradient: z rows, tile-major
        double* yt = nullptr;                              // gradient of tree states: y rows, tile-major
    };
    std::mutex mu;
    std::vector<Set> spare;                                // zeroed, unassigned
    // key: (capture id + 1 while the stream is being captured | 0 for eager launches | a per-thread tag for hipStreamPerThread,
    // which is ONE handle value for as many streams as there are threads), stream
    std::map<std::pair<unsigned long long, hipStream_t>, Set> assigned;
    std::vector<Set> retired;                              // sets a timing probe left with tickets taken and slots unread: never reused
    hipStream_t init_stream = nullptr;
};

static hipError_t new_set(SplitHost* s, SplitHost::Set& out)
{
    // may run while the calling thread captures a stream: allocations are legal under the relaxed capture mode, and the
    // clearing memset goes to a stream of the pool's own
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    hipError_t e = hipMalloc((void**)&out.partials, kSplitScratchDoubles * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&out.counters, kSplitCounters * sizeof(unsigned));
    // zt / yt (2 x 1024 chains x 16 NB rows: 16 MiB each at N = 1024) only when a gradient call needs them: scratch_for(.., grad)
    if (e == hipSuccess) e = hipMemsetAsync(out.counters, 0, kSplitCounters * sizeof(unsigned), s->init_stream);
    if (e == hipSuccess) {                                 // every slot starts as "no partial sum yet"
        static_assert(sizeof(unsigned long long) == sizeof(double), "");
        hipLaunchKernelGGL(k_split_poison, dim3((unsigned)((kSplitScratchDoubles + 255) / 256)), dim3(256), 0, s->init_stream,
                           reinterpret_cast<unsigned long long*>(out.partials), (size_t)kSplitScratchDoubles);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s->init_stream);
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    return e;
}

template <int NC, bool TREE>
static hipError_t split_allow_lds()
{
    if (split_lds_bytes<NC>() <= 64 * 1024) return hipSuccess;
    if (hipError_t e = hipFuncSetAttribute((const void*)k_split<NC, TREE, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)split_lds_bytes<NC>())) return e;
    if (hipError_t e = hipFuncSetAttribute((const void*)k_split<NC, TREE, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)split_lds_bytes<NC>())) return e;
    if (!TREE) return hipFuncSetAttribute((const void*)k_split<NC, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)split_lds_bytes<NC>());
    return hipSuccess;
}

SplitHost* split_host_create(int n, const double* W_rowmajor, hipError_t* err)
{
    *err = hipSuccess;
    SplitHost* s = new SplitHost();
    s->n = n;
    auto fail = [&](hipError_t e) {
        *err = e;
        split_host_destroy(s);
        return (SplitHost*)nullptr;
    };
    if (hipError_t e = hipGetDevice(&s->device)) return fail(e);
    if (hipError_t e = hipStreamCreateWithFlags(&s->init_stream, hipStreamNonBlocking)) return fail(e);
    const int NB = (n + 15) / 16;
    s->zt_doubles = (size_t)kSplitMaxBatch * 16 * NB;
    const std::vector<double> W(W_rowmajor, W_rowmajor + (size_t)n * n);
    std::vector<double> Wb((size_t)n * n, 0.0);            // J W^T J: element (i, j) = W(n - 1 - j, n - 1 - i), lower triangular again
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) Wb[(size_t)i * n + j] = W[(size_t)(n - 1 - j) * n + (n - 1 - i)];
    const int Gs[3] = {8, 16, 32};
    for (int v = 0; v < 3; ++v) {
        const int G = Gs[v];
        if (v > 0 && (int64_t)NB * (NB + 1) < 32 * (int64_t)G) continue;   // fewer than 16 tiles per wave: not worth a variant
        for (int back = 0; back < 2; ++back) {
            SplitScheduleHost h;
            build_split_schedule(n, G, back ? Wb : W, h);
            if (h.nc > 4) continue;
            void* dW = nullptr;
            if (hipError_t e = hipMalloc(&dW, h.Ws.size() * sizeof(double))) return fail(e);
            s->dev_allocs.push_back(dW);
            if (hipError_t e = hipMemcpy(dW, h.Ws.data(), h.Ws.size() * sizeof(double), hipMemcpyHostToDevice)) return fail(e);
            SplitSched& S = back ? s->sched_b[v] : s->sched[v];
            S.G = G;
            S.nc = h.nc;
            S.NB = h.NB;
            S.Ws = (const double*)dW;
            for (int g = 0; g < G; ++g) {
                S.base[g] = h.base[g];
                const SpGroup q = sp_group(h.NB, G, g);
                int ncut = 0;
                sp_for_each_cut(h.NB, G, g, q, [&](int wf, int wl, int blk) {
                    if (ncut < SP_NW - 1) S.cuts[g][1 + ncut] = wf | (wl << 4) | (blk << 8);
                    ++ncut;
                });
                if (ncut > SP_NW - 1) return fail(hipErrorInvalidValue);     // (a run is contiguous: at most SP_NW - 1 cuts)
                S.cuts[g][0] = ncut;
            }
        }
    }
    // more than 64 KiB of dynamic LDS has to be allowed once per kernel and device, outside any stream capture
    if (hipError_t e = split_allow_lds<2, false>()) return fail(e);
    if (hipError_t e = split_allow_lds<2, true>()) return fail(e);
    if (hipError_t e = split_allow_lds<3, false>()) return fail(e);
    if (hipError_t e = split_allow_lds<3, true>()) return fail(e);
    if (hipError_t e = split_allow_lds<4, false>()) return fail(e);
    if (hipError_t e = split_allow_lds<4, true>()) return fail(e);
    for (int i = 0; i < 4; ++i) {                          // a few scratch sets ahead, so that a first use under capture finds one
        SplitHost::Set set;
        if (hipError_t e = new_set(s, set)) return fail(e);
        s->spare.push_back(set);
    }
    return s;
}

void split_host_destroy(SplitHost* s)
{
    if (!s) return;
    for (void* p : s->dev_allocs) (void)hipFree(p);
    auto free_set = [](SplitHost::Set& x) {
        if (x.partials) (void)hipFree(x.partials);
        if (x.counters) (void)hipFree(x.counters);
        if (x.zt) (void)hipFree(x.zt);
        if (x.yt) (void)hipFree(x.yt);
    };
    for (auto& x : s->spare) free_set(x);
    for (auto& kv : s->assigned) free_set(kv.second);
    for (auto& x : s->retired) free_set(x);
    if (s->init_stream) (void)hipStreamDestroy(s->init_stream);
    delete s;
}

// The scratch set of a launch on `st`: launches on one stream are ordered, so they share a set; a stream being captured gets
// a set of the capture's own (the capture id), because the graph may later be replayed on any stream, concurrently with eager
// launches on the stream it was captured from.
static std::pair<unsigned long long, hipStream_t> scratch_key(hipStream_t st, hipStreamCaptureStatus status, unsigned long long id)
{
    if (status == hipStreamCaptureStatusActive) return {id + 1, st};
    if (st == hipStreamPerThread)                          // distinct streams behind one sentinel value: key them by thread
        return {0x8000000000000000ull | (unsigned long long)std::hash<std::thread::id>()(std::this_thread::get_id()), st};
    return {0ull, st};
}

static hipError_t scratch_for(SplitHost* s, hipStream_t st, bool grad, SplitHost::Set& out)
{
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipError_t e = hipStreamGetCaptureInfo(st, &status, &id)) return e;
    if (status == hipStreamCaptureStatusInvalidated) return hipErrorStreamCaptureInvalidated;
    const auto key = scratch_key(st, status, id);
    std::lock_guard<std::mutex> lock(s->mu);
    auto it = s->assigned.find(key);
    if (it == s->assigned.end()) {
        SplitHost::Set set;
        if (!s->spare.empty()) {
            set = s->spare.back();
            s->spare.pop_back();
        } else if (hipError_t e = new_set(s, set)) {
            return e;
        }
        it = s->assigned.emplace(key, set).first;
    }
    if (grad && it->second.zt == nullptr) {                // first gradient call on this stream / capture
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        hipError_t e = hipMalloc((void**)&it->second.zt, s->zt_doubles * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void**)&it->second.yt, s->zt_doubles * sizeof(double));
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        if (e != hipSuccess) return e;
    }
    out = it->second;
    return hipSuccess;
}

// A timing probe (MCD_SPLIT_PROBE) ends the kernel with tickets taken and slots unread: the set is retired, the next launch
// on the stream draws a clean one.
static void scratch_retire(SplitHost* s, hipStream_t st)
{
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo(st, &status, &id) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(s->mu);
    auto it = s->assigned.find(scratch_key(st, status, id));
    if (it == s->assigned.end()) return;
    s->retired.push_back(it->second);
    s->assigned.erase(it);
}

// mcd_mvn_release_stream: the caller is done with `st` (it is idle or about to be destroyed) -- its eager set goes back to the
// pool.  The sets of captures stay: a graph instantiated from the capture may still be replayed.
hipError_t split_release_stream(SplitHost* s, hipStream_t st)
{
    if (!s) return hipSuccess;
    if (hipError_t e = hipStreamSynchronize(st)) return e;  // the last launch's reader has re-poisoned its slots and reset the counter
    std::lock_guard<std::mutex> lock(s->mu);
    auto it = s->assigned.find(scratch_key(st, hipStreamCaptureStatusNone, 0));
    if (it != s->assigned.end()) {
        s->spare.push_back(it->second);
        s->assigned.erase(it);
    }
    return hipSuccess;
}

static int pick_variant(const SplitHost* s, int64_t batch)
{
    const int force = opt_or(OPT_SPLIT_G, 0);              // tuning / tests (mcd_set_option "MCD_SPLIT_G"): force G
    const int64_t tiles = (batch + 15) / 16;
    int best = 0;
    for (int v = 0; v < 3; ++v) {
        if (s->sched[v].G == 0) continue;
        if (force == s->sched[v].G) return v;
        if (tiles * s->sched[v].G <= 256) best = v;        // more row groups while every workgroup still has a CU to itself
    }
    return best;
}

// MODE 0: ll.  MODE 1: ll + gradient: the z pass, the y pass (schedule of J W^T J) and, for tree states, the chain rule.
template <bool TREE, int MODE>
static hipError_t launch_split(const MvnDev& M, const WideSrc& A, int64_t batch, double* ll, double* G, int64_t ldg, double* gR, double* gtH,
                               double* grMu, hipStream_t st, bool z_only = false, const double** zt_out = nullptr, int* nr_out = nullptr)
{
    if (batch <= 0) return hipSuccess;
    SplitHost* s = const_cast<SplitHost*>(M.split);
    if (s == nullptr || batch > kSplitMaxBatch) return hipErrorInvalidValue;
    const int v = pick_variant(s, batch);
    const SplitSched& S = s->sched[v];
    if (S.G == 0) return hipErrorInvalidValue;
    SplitHost::Set set;
    if (hipError_t e = scratch_for(s, st, MODE == 1, set)) return e;
    int scatter = opt_or(OPT_SPLIT_SCATTER, 0) & 1;        // tests (mcd_set_option "MCD_SPLIT_SCATTER"): a tile's row groups on different XCDs
    if (opt_or(OPT_SPLIT_NOROT, 0)) scatter |= 2;
    scatter |= opt_or(OPT_SPLIT_PROBE, 0) & 28;            // timing probes (results are then garbage): 4 launch only, 8 + staging, 16 all but the hand-over
    const bool probe = (scatter & 28) != 0;

    const int64_t tiles = (batch + 15) / 16;
    const unsigned grid = (scatter & 1) ? (unsigned)(tiles * S.G) : (unsigned)(((tiles + 7) / 8) * 8 * S.G);
    SplitGrad Gr{set.zt, nullptr, 0, 0};
#define MCD_SPLIT_LAUNCH(NC_, TREE_, MODE_, S_) \
    hipLaunchKernelGGL((k_split<NC_, TREE_, MODE_>), dim3(grid), dim3(tiles * S.G <= 256 ? split_threads<NC_>() : 64 * SP_NW), split_lds_bytes<NC_>(), st, M, S_, A, Gr, batch, scatter, ll, set.partials, set.counters)
    switch (S.nc) {
    case 1: MCD_SPLIT_LAUNCH(1, TREE, MODE, S); break;
    case 2: MCD_SPLIT_LAUNCH(2, TREE, MODE, S); break;
    case 3: MCD_SPLIT_LAUNCH(3, TREE, MODE, S); break;
    case 4: MCD_SPLIT_LAUNCH(4, TREE, MODE, S); break;
    default: return hipErrorInvalidValue;
    }
    if (hipError_t e = hipGetLastError()) return e;
    if (zt_out) *zt_out = set.zt;
    if (nr_out) *nr_out = 16 * S.NB;
    if constexpr (MODE == 1) {
        if (z_only) {                                        // ll and the z tiles only (the incremental Metropolis-Hastings path keeps z)
            if (probe) scratch_retire(s, st);
            return hipGetLastError();
        }
        const SplitSched& Sb = s->sched_b[v];
        if (Sb.G != S.G || Sb.nc != S.nc) return hipErrorInvalidValue;
        Gr.out = TREE ? set.yt : G;
        Gr.ldo = ldg;
        Gr.t = TREE ? 1 : 0;
        switch (S.nc) {
        case 1: MCD_SPLIT_LAUNCH(1, false, 2, Sb); break;
        case 2: MCD_SPLIT_LAUNCH(2, false, 2, Sb); break;
        case 3: MCD_SPLIT_LAUNCH(3, false, 2, Sb); break;
        case 4: MCD_SPLIT_LAUNCH(4, false, 2, Sb); break;
        default: return hipErrorInvalidValue;
        }
        if (hipError_t e = hipGetLastError()) return e;
        if constexpr (TREE) {
            const size_t lds_bytes = (size_t)(A.T.n_nodes_pad + 16) * sizeof(double);
            if (lds_bytes > 64 * 1024) return hipErrorInvalidValue;
            const int nthr = A.T.n_nodes > 512 ? 1024 : A.T.n_nodes > 256 ? 512 : 256;
            hipLaunchKernelGGL(k_split_tree_chain, dim3((unsigned)batch), dim3(nthr), lds_bytes, st, M.n, 16 * S.NB, A.T, A.H, A.Rt, A.lds, A.tH, A.rMu,
                               (const double*)set.yt, G, gR, gtH, grMu);
        }
    }
#undef MCD_SPLIT_LAUNCH
    if (probe) scratch_retire(s, st);
    return hipGetLastError();
}

hipError_t launch_logpdf_split(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st)
{
    WideSrc A{};
    A.X = X;
    A.ldx = ldx;
    return launch_split<false, 0>(M, A, batch, ll, nullptr, 0, nullptr, nullptr, nullptr, st);
}

// ll, and z = L^-1 (x - mu) left tile-major in the stream's scratch set: element (row r, chain b) at zt[((b / 16) nr + r) 16 + b % 16];
// valid until the next gradient-type launch on the stream (k_mh_inc.hip copies what it keeps)
hipError_t launch_logpdf_split_z(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, const double** zt, int* nr, hipStream_t st)
{
    WideSrc A{};
    A.X = X;
    A.ldx = ldx;
    return launch_split<false, 1>(M, A, batch, ll, nullptr, 0, nullptr, nullptr, nullptr, st, true, zt, nr);
}

// ll and d ll / d x = -Sigma^-1 (x - mu) as two triangular products on the row-split schedule (G may be X: the first launch
// has read every x before the second writes)
hipError_t launch_grad_split(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, double* G, int64_t ldg, hipStream_t st)
{
    WideSrc A{};
    A.X = X;
    A.ldx = ldx;
    return launch_split<false, 1>(M, A, batch, ll, G, ldg, nullptr, nullptr, nullptr, st);
}

hipError_t launch_tree_logpdf_split(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                    const double* rMu, int64_t batch, double* ll, double* logjac, hipStream_t st)
{
    WideSrc A{};
    A.T = T;
    A.H = H;
    A.Rt = Rt;
    A.lds = lds;
    A.tH = tH;
    A.rMu = rMu;
    A.logjac = logjac;
    return launch_split<true, 0>(M, A, batch, ll, nullptr, 0, nullptr, nullptr, nullptr, st);
}

hipError_t launch_tree_grad_split(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                  const double* rMu, int64_t batch, double* ll, double* gH, double* gR, double* gtH, double* grMu, hipStream_t st)
{
    WideSrc A{};
    A.T = T;
    A.H = H;
    A.Rt = Rt;
    A.lds = lds;
    A.tH = tH;
    A.rMu = rMu;
    return launch_split<true, 1>(M, A, batch, ll, gH, lds, gR, gtH, grMu, st);
}

}  // namespace mcd

#ifdef MCD_SPLIT_STAMP
extern "C" int mcd_split_debug_stamps(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mcd::g_split_dbg), sizeof(unsigned long long) * 2 * mcd::SP_NW * 16);
}
#endif
