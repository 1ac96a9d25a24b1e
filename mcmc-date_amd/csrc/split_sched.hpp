// split_sched.hpp -- the schedule of the row-split form (k_split.hip) as arithmetic, shared by the host (host_factor.cpp packs
// the tile runs with it and checks itself against it) and the device (the kernel derives its runs from blockIdx alone: no
// table has to be fetched before the first tile can be requested).
//
// W = L^-1 has NB = ceil(N / 16) row blocks of 16 rows; block b holds the k tiles 0 .. 4 (b + 1) - 1 (4 columns each).
// G row groups per 16-chain tile.  Round r deals the blocks r G .. r G + G - 1 to the groups forwards (r even) or backwards
// (r odd), so every group gets small and large blocks alike.  A group's tiles -- block after block in ascending order, k
// ascending inside a block -- form one stream of Tg tiles = Tg / 4 groups of 4, dealt to the SP_NW waves as evenly as whole
// groups allow (the first `rem` waves take one group more): one contiguous run per wave.
// Inside a run, the tiles of one block are a SEGMENT {first k tile, tiles, kind}: kind 0 = the wave holds the whole block
// and squares the z tile itself; otherwise the block is cut between waves and the wave's partial z tile goes to an LDS slot:
// kind 1 = the run begins inside the block (slot 2 w - 1), kind 2 = the run begins with the block but ends inside it (slot
// 2 w).  A run is contiguous, so only its first and its last segment can be partial: 2 SP_NW - 2 slots.
#pragma once

#if defined(__HIPCC__)
#define SP_HD __host__ __device__ inline
#else
#define SP_HD inline
#endif

namespace mcd {

SP_HD int sp_block(int G, int g, int r) { return r * G + ((r & 1) ? G - 1 - g : g); }

constexpr int SP_NW = 8;                  // waves per workgroup = runs per group
constexpr int SP_NSLOT = 2 * SP_NW - 2;   // LDS slots for partial z tiles

struct SpGroup {
    int Tg;      // tiles of the group
    int base;    // groups of 4 tiles per wave: the first `rem` waves take base + 1, the others base
    int rem;
    int ncols;   // residual columns the group needs: 16 (last block + 1)
};

SP_HD SpGroup sp_group(int NB, int G, int g)
{
    SpGroup q{0, 0, 0, 0};
    for (int r = 0; r * G < NB; ++r) {
        const int b = sp_block(G, g, r);
        if (b >= NB) continue;                            // only the last round can be short
        q.Tg += 4 * (b + 1);
        q.ncols = 16 * (b + 1);
    }
    q.base = (q.Tg >> 2) / SP_NW;
    q.rem = (q.Tg >> 2) - q.base * SP_NW;
    return q;
}

// first tile of wave w's run (w = SP_NW: the end of the stream)
SP_HD int sp_run_start(const SpGroup& q, int w) { return 4 * (w * q.base + (w < q.rem ? w : q.rem)); }

// the wave whose run holds stream position pos (0 <= pos < Tg); no division: the device has none
SP_HD int sp_wave_of(const SpGroup& q, int pos)
{
    int w = 0;
#pragma unroll
    for (int x = 1; x < SP_NW; ++x) w += (sp_run_start(q, x) <= pos && sp_run_start(q, x) < q.Tg) ? 1 : 0;
    return w;
}

// f(k0, nt, kind) for every segment of wave `wave` of group `g`, in stream order
template <class F>
SP_HD void sp_for_each_segment(int NB, int G, int g, const SpGroup& q, int wave, F&& f)
{
    const int lo = sp_run_start(q, wave), hi = sp_run_start(q, wave + 1);
    int s = 0;
    for (int r = 0; r * G < NB; ++r) {
        const int b = sp_block(G, g, r);
        if (b >= NB) continue;
        const int nt = 4 * (b + 1);
        const int a = lo > s ? lo : s, e = hi < s + nt ? hi : s + nt;
        if (e > a) f(a - s, e - a, (a == s && e == s + nt) ? 0 : (a > s ? 1 : 2));
        s += nt;
    }
}

// f(wf, wl, b) for every block b of group `g` that is cut between waves: its partial z tiles sit in the slot 2 wf of wave wf (the
// block ends that wave's run) and in the slots 2 w - 1 of the waves wf < w <= wl (their runs begin inside it); wave order
template <class F>
SP_HD void sp_for_each_cut(int NB, int G, int g, const SpGroup& q, F&& f)
{
    int s = 0;
    for (int r = 0; r * G < NB; ++r) {
        const int b = sp_block(G, g, r);
        if (b >= NB) continue;
        const int e = s + 4 * (b + 1);
        const int wf = sp_wave_of(q, s), wl = sp_wave_of(q, e - 1);
        if (wl > wf) f(wf, wl, b);
        s = e;
    }
}

}  // namespace mcd
