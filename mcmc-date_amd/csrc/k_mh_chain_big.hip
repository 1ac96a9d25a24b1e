// k_mh_chain_big.hip -- a whole Metropolis-Hastings-Green schedule in ONE launch for trees of 65 .. 514 nodes at a sampler's
// batch (gfx950).  SURVEY.md 8(f) row f2; the persistent form VERDICT (round 1, item 5) asked for, with the factor streamed
// instead of resident: it does not fit (263 KB at N = 255), but it stays in L2, every workgroup streams it at the same time, and
// what a launch per step really costs -- 1.7 us of dispatch twice per step, every chain's state, posterior terms, tuning
// parameter and draws fetched from memory at the start of each launch and written back at its end -- disappears.
//
// A workgroup owns TWO chains for the complete schedule: two chain waves and two loader waves, exactly the geometry of the
// sweep's tree-likelihood launch at this batch size (k_tree_logpdf<R, 1, 2, 2>).  Per step, with no launch in between:
//   chain wave    propose (mh_device.hpp) -> changed blocks of the ln prior (prior_device.hpp) -> distances of the proposed state
//                 from LDS -> forward sweep against the LDS ring (mvn_device.hpp: fwd_compute) -> accept / reject
//   loader wave   streams the factor of Sigma through the ring once per step (fwd_loader), in step with the chain waves
//                 through the ring's workgroup barriers
// A chain's state (heights, rates, the five scalars, the three blocks of its ln prior, ln likelihood, ln root-branch
// Jacobian), its tuning parameters and counters live in LDS and registers from the first step to the last; so do the tree tables
// the proposals and the prior look up (parent, sub tree size, children).
//
// The arithmetic is the two-launch path's (k_mh.hip + k_tree_logpdf.hip): the same proposal and prior functions on the same
// numbers, the same sweep templates with the same R, the same reduction -- a chain advanced by this kernel is bit-identical to
// the same chain advanced by that path (tests/test_gpu_mh.py::test_streaming_chain_kernel_equals_two_launch_path).  Steps whose
// proposal cannot move the likelihood (birth rate, death rate, rate variance) skip the sweep on both sides: the decision is a
// function of the proposal row alone, so chain waves and loader waves take it alike.
//
// Round 3: most proposals of the cycle move at most a handful of branch distances (slide a node: three; scale one branch rate:
// one; small sub trees), and the full sweep was 43 % of a step.  The chain wave keeps z = L^-1 (d - mu) of the CURRENT state in
// registers; for a proposal whose table row is marked sparse (mcd_mh_create: MhDev::sparse) it forms z' = z + sum_j delta_j W[:, j]
// over the distances that actually moved (found by comparing d' with d: right for any number of them) with columns of W = L^-1
// from L2 (MvnDev::Wc, 2 KiB each), requested before the ln prior is evaluated and used after it, and q' = |z'|^2; the loader waves
// do not stream on such steps.  Dense proposals take the sweep as before, which also leaves z' exact; every 256 steps (and at the
// start of a launch) z is recomputed by a sweep of the current state.  ln likelihood values then agree with a full evaluation
// to rounding (1e-13 relative), not bit for bit: decisions, states and counters still equal the two-launch path's, the traced
// ln acceptance ratios and the ln likelihood agree within the twin's tolerance (tests/test_gpu_mh.py).
//
// Reference: the loop this replaces is `mhg`'s iteration of `mcmc` [external] driven from app/Main.hs:460-479 with the cycle of
// app/Definitions.hs:256-278; likelihood app/Probability.hs:166-173, 195-207; jacobianRootBranch :393-410.
#include "mvn_device.hpp"
#include "options.h"
#include "mh_device.hpp"
#include "prior_device.hpp"

#include <atomic>

namespace mcd {

// does a proposal of this row move the distances?  (MCD_PROP_SCALE_SCALAR on birth rate, death rate or rate variance does not)
__device__ __forceinline__ bool mhb_moves_likelihood(int kind, int node)
{
    return !(kind == MCD_PROP_SCALE_SCALAR && (node == 0 || node == 1 || node == 4));
}

// LDS per chain (doubles): 4 state rows; the per-node summands of the birth-death and the clock block of the ln prior, current and
// proposed, with the step that wrote the proposed one (4 n_nodes doubles + 2 n_nodes int32).  The tuning parameters and the
// accept / try counters ([n_prop] each: 22 KiB per chain at 385 nodes) stay in global memory -- a step reads one tuning parameter and
// bumps two counters -- which is what lets trees of up to 514 nodes (N <= 512, R = 6 and 8) fit beside the 64-KiB ring.
__host__ __device__ inline size_t mhb_chain_doubles(int n_nodes, int /*n_prop*/) { return 9 * (size_t)n_nodes + 1; }
// ... and per workgroup: the tree tables (five int32 arrays of n_nodes, rounded up to doubles)
__host__ __device__ inline size_t mhb_table_doubles(int n_nodes) { return (5 * (size_t)n_nodes + 1) / 2 + 1; }

constexpr int kMhbRefresh = 256;   // steps between two recomputations of z by a full sweep of the current state (a power of two)

// R <= 4 (up to 258 nodes): on the steps whose proposal is sparse the loader waves have nothing to stream, and the chain wave spent a
// quarter of such a step on the likelihood's side (all distances, which of them moved, their columns, z').  There the loader wave of a
// chain acts as its LIKELIHOOD wave, as in k_mh_segment.hip: it derives the moved distance slots from the nodes the proposal writes,
// forms z' = z + sum_j delta_j W[:, j] and hands |z'|^2 back while the chain wave evaluates the ln prior.  The current distances and z
// live in LDS (both waves need them: dense proposals and the refreshes are still swept by the chain wave); a few LDS words carry the
// hand-over (request, reply, decision, and `done`: the likelihood wave has committed, the chain wave may touch distances and z).
template <int R> struct MhbHelp { static constexpr bool on = (R <= 4); };
constexpr int kMhbList = 64;       // moved distances of one sparse proposal at most (mh_capi.cpp: kMhSparseSlots <= this)
struct MhbWords {
    int req, resp, dec, done;      // step + 1 | step + 1 | 2 (step + 1) + accepted | step + 1
    int cnt, have0;                // moved distances (-1: more than the list holds); slot 0 among them
    double q, lj, s1;              // |z'|^2; ln jacobianRootBranch of the proposal (have0); tH * rMu of the proposal
    double pad[2];
};
static_assert(sizeof(MhbWords) == 64, "eight doubles of LDS");
constexpr int kMhbApplyDoubles = (int)((sizeof(PropApply) + 7) / 8);
// LDS of the likelihood-wave form, in doubles: per chain the current distances and z [np], the list (new distance, delta, slot),
// the slots' marks, the words, the proposal's transform; shared: slot -> node, slot -> parent, node -> slot (int16)
__host__ __device__ inline size_t mhb_help_chain_doubles(int np)
{
    return 2 * (size_t)np + 2 * (size_t)kMhbList + (size_t)kMhbList / 2 + (size_t)np / 2 + 8 + (size_t)kMhbApplyDoubles;
}
__host__ __device__ inline size_t mhb_help_table_doubles(int n_nodes, int np) { return ((size_t)n_nodes + 2 * (size_t)np + 3) / 4 + 1; }
// One pass of the loader waves over the factor, as a function of its own: inside the kernel its staging registers were allocated
// together with the chain role's code and spilled from R = 6 up (76 / 674 registers at R = 6 / 8); a call costs a few hundred cycles
// per pass of some 14 000.
__host__ __device__ inline size_t mhb_lds_bytes(int n_nodes, int n_prop, int R)
{
    size_t d = mhb_table_doubles(n_nodes) + 2 * mhb_chain_doubles(n_nodes, n_prop);
    if (R <= 4) d += mhb_help_table_doubles(n_nodes, 64 * R) + 2 * mhb_help_chain_doubles(64 * R);      // (MhbHelp<R>::on)
    return sizeof(double) * d;
}
// ... and, where it still fits beside the 64-KiB ring, the calibration and constraint tables (prior_device.hpp: prior_stage_node_tables)
__host__ __device__ inline size_t mhb_node_tables_bytes(int n_nodes, int n_prop, int R, int n_cal, int n_con)
{
    const size_t need = sizeof(double) * prior_node_tables_doubles(n_cal, n_con);
    return (need > 0 && mhb_lds_bytes(n_nodes, n_prop, R) + need <= 94 * 1024) ? need : 0;
}

template <int R, int LW>
__device__ __forceinline__ void mhb_stream_pass_body(const double* __restrict__ Ft, d2* ring, int lw, int lane, int ncols)
{
    MCD_ACC_DECL
    Stage<R, LW> st;
    // chunk 0 into the ring, chunks 1 and 2 requested -- all of it while the chain waves propose and evaluate the prior
    // (the stand-alone launch requests 1 and 2 after the barrier, out of the way of the compute waves' state loads;
    // here nothing competes and the stream's first round trip would be exposed at every step)
    fwd_loader_prologue<R, LW>(Ft, ring, st, lw, lane);
    fwd_loader_start<R, LW>(Ft, st, lw, lane);
    lds_barrier();
    fwd_loader<R, LW, 0>(Ft, ring, st, lw, lane, ncols MCD_ACC_ARGS);
}
template <int R, int LW>
__device__ __noinline__ void mhb_stream_pass_call(const double* __restrict__ Ft, d2* ring, int lw, int lane, int ncols)
{
    mhb_stream_pass_body<R, LW>((const double*)(const __attribute__((address_space(1))) double*)Ft, ring, lw, lane, ncols);   // (global memory: global loads)
}
// (R <= 4 keeps the body inline: no call frame, no scratch at all in those instantiations)
template <int R, int LW>
__device__ __forceinline__ void mhb_stream_pass(const double* __restrict__ Ft, d2* ring, int lw, int lane, int ncols)
{
    if constexpr (R <= 4)
        mhb_stream_pass_body<R, LW>(Ft, ring, lw, lane, ncols);
    else
        mhb_stream_pass_call<R, LW>(Ft, ring, lw, lane, ncols);
}

template <int R> struct MhbCols { static constexpr int N = (R <= 4) ? 4 : 2; };   // columns of L^-1 in flight per batch of moved distances

template <int R>
__global__ __launch_bounds__(256, 1) void k_mh_chain_big(MhDev M, MvnDev V, TreeDev T, PriorDev P, const int32_t* __restrict__ sched,
                                                      int64_t n_steps, int32_t S, int accumulate, uint64_t step0, uint64_t seed,
                                                      double* __restrict__ trace_alpha, int8_t* __restrict__ trace_accept)
{
    constexpr int CW = 2, LW = Cfg<R>::LW;
    static_assert(LW == 2, "geometry of k_tree_logpdf<R, 1, 2, 2>");
    __shared__ d2 ring[2 * Cfg<R>::SU * 64];
    extern __shared__ double dyn[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nn = M.n_nodes, NP = M.n_prop;
    const int ncols = V.ncols;
    MCD_ACC_DECL
    constexpr bool HLP = MhbHelp<R>::on;
    constexpr int NPadC = 64 * R;
    // (HLP) the likelihood-wave form's LDS, behind the tables and the two chains' arrays
    const int cs_ = (wave >= CW) ? wave - CW : wave;         // the chain of the workgroup this wave works for
    int16_t* ts_node = reinterpret_cast<int16_t*>(dyn + mhb_table_doubles(nn) + 2 * mhb_chain_doubles(nn, NP));
    int16_t* ts_parent = ts_node + NPadC;
    int16_t* ts_of = ts_parent + NPadC;
    double* hc0 = dyn + mhb_table_doubles(nn) + 2 * mhb_chain_doubles(nn, NP) + mhb_help_table_doubles(nn, NPadC) + (size_t)cs_ * mhb_help_chain_doubles(NPadC);
    double* dcur_l = hc0;                                    // [NPad] distances of the current state
    double* z_l = dcur_l + NPadC;                            // [NPad] z = L^-1 (d - mu) of the current state
    double* l_dnew = z_l + NPadC;
    double* l_delta = l_dnew + kMhbList;
    int32_t* l_j = reinterpret_cast<int32_t*>(l_delta + kMhbList);
    int32_t* mark = l_j + kMhbList;
    MhbWords* words = reinterpret_cast<MhbWords*>(reinterpret_cast<double*>(mark + NPadC));
    PropApply* A_lds = reinterpret_cast<PropApply*>(reinterpret_cast<double*>(words) + 8);
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    auto poll = [&](int* wp, int want, int shift) -> int {   // (mh_device.hpp: the words typed as LDS, fences that wait for LDS only)
        lds_vint_t* w = lds_vint(wp);
        int v = *w;
        while ((v >> shift) != want) {
            __builtin_amdgcn_s_sleep(1);
            v = *w;
        }
        lds_acquire_fence();
        return v;
    };
    auto post = [&](int* wp, int value) {
        lds_publish_fence();
        *lds_vint(wp) = value;
    };

    // ---- loader waves: one pass over the factor per step that needs the likelihood
    if (wave >= CW) {
        const int lw = wave - CW;
        int p = sched[0];
        const int vz = mh_vzero();                           // (mh_device.hpp: what travels ahead is loaded by vector loads)
        int p_next = sched[(n_steps > 1 ? 1 : 0) + vz];      // (the schedule's entries two steps ahead: a row's loads need its index)
        int kind = M.kind[p], node = M.node[p];
        const bool inc = V.Wc != nullptr;
        int sp = inc ? M.sparse[p] : 0;
        // (the factor is in global memory: said so, the stream's loads are global loads -- as generic pointer they compiled to flat loads,
        // which count against the LDS counter as well and so tie the ring's writes to the whole stream)
        const double* Ft_g = (const double*)(const __attribute__((address_space(1))) double*)V.Ft;
        // (HLP) this wave's chain: its proposed state, and the tables the chain waves put in LDS before their first barrier
        const int32_t* tb_first_ = reinterpret_cast<const int32_t*>(dyn) + 2 * nn;
        const int32_t* tb_nch_ = tb_first_ + nn;
        const int32_t* tb_second_ = tb_nch_ + nn;
        const double* Hp_ = dyn + mhb_table_doubles(nn) + (size_t)lw * mhb_chain_doubles(nn, NP) + 2 * (size_t)nn;
        const double* Rp_ = Hp_ + nn;
        const int rr_ = T.root_right;
        MhDev Mt_ = M;                                       // (mh_propose_ranges reads the sub tree sizes -- the chain waves' table in LDS -- and the braces' pointers)
        Mt_.size = reinterpret_cast<const int32_t*>(dyn) + nn;
        for (int64_t gs = 0; gs < n_steps; ++gs) {
            const int p_next2 = sched[((gs + 2 < n_steps) ? gs + 2 : gs) + vz];
            const int kind_next = M.kind[p_next], node_next = M.node[p_next];     // (travel while this step streams)
            const int sp_next = inc ? M.sparse[p_next] : 0;
            // z of the current state (start of the launch, then every 256 steps), then the step's own sweep if its proposal is dense
            // (one call site: the stream is a long unrolled body)
            const int passes = ((inc && (gs & (kMhbRefresh - 1)) == 0) ? 1 : 0) + ((mhb_moves_likelihood(kind, node) && !sp) ? 1 : 0);
            for (int r = 0; r < passes; ++r) mhb_stream_pass<R, LW>(Ft_g, ring, lw, lane, ncols);
            if constexpr (HLP) {
                if (inc && sp && mhb_moves_likelihood(kind, node)) {
                    // ---- the likelihood wave of a sparse step (k_mh_segment.hip has the same: the slots the written nodes feed, each once)
                    const int tag = (int)gs + 1;
                    // AHEAD of the request, while the chain wave draws the proposal: which nodes the proposal writes follows from its table row
                    // and the topology alone (mh_propose_ranges) -- hence the list of moved slots, and the first four columns of L^-1 are
                    // requested at once.  The guess is compared with the transform the proposal posts; the rare mismatch (a proposal that
                    // bails out on an invalid state) lists again, with a tag of its own.  (The transform's integer fields travel as scalars:
                    // a struct passed around here ends up in scratch memory.)
                    struct MhbRanges {
                        int kind, hlo, hhi, hlo2, hhi2, rlo, rhi, pt1, pt2, rp1, rp2, rp3, brace_lo, brace_hi;
                    };
                    auto build_list = [&](int a_kind, int a_hlo, int a_hhi, int a_hlo2, int a_hhi2, int a_rlo, int a_rhi, int a_pt1, int a_pt2, int a_rp1, int a_rp2, int a_rp3, int a_brace_lo, int a_brace_hi, int tagf) __attribute__((always_inline)) -> int {
                        const MhbRanges A{a_kind, a_hlo, a_hhi, a_hlo2, a_hhi2, a_rlo, a_rhi, a_pt1, a_pt2, a_rp1, a_rp2, a_rp3, a_brace_lo, a_brace_hi};
                        int cnt = 0;
                        auto emit = [&](bool active, int node_) {
                            if (__builtin_amdgcn_ballot_w64(active) == 0) return;
                            const int slot = active ? (int)ts_of[node_] : -1;
                            bool mine = false;
                            if (slot >= 0) mine = atomicMax(&mark[slot], tagf) != tagf;      // (the tags count upwards within a launch)
                            const uint64_t mk = __builtin_amdgcn_ballot_w64(mine);
                            if (mine) {
                                const int pos = cnt + (int)__builtin_popcountll(mk & lt_mask);
                                if (pos < kMhbList) l_j[pos] = slot;
                            }
                            cnt += (int)__builtin_popcountll(mk);
                        };
                        auto emit_height = [&](bool active, int w) {
                            emit(active, w);
                            const int nc = active ? tb_nch_[w] : 0;
                            emit(nc > 0, active ? tb_first_[w] : 0);
                            emit(nc > 1, active ? tb_second_[w] : 0);
                        };
                        for (int w0 = A.hlo; w0 < A.hhi; w0 += 64) emit_height(w0 + lane < A.hhi, w0 + lane);
                        for (int w0 = A.hlo2; w0 < A.hhi2; w0 += 64) emit_height(w0 + lane < A.hhi2, w0 + lane);
                        for (int w0 = A.rlo; w0 < A.rhi; w0 += 64) emit(w0 + lane < A.rhi, w0 + lane);
                        {
                            const int g = lane / 3, r = lane - 3 * g;
                            int cand = -1;
                            if (lane < 6) {
                                const int base = (g == 0) ? A.pt1 : A.pt2;
                                if (base >= 0) cand = (r == 0) ? base : (tb_nch_[base] >= r) ? (r == 1 ? tb_first_[base] : tb_second_[base]) : -1;
                            } else if (lane < 9) {
                                cand = (r == 0) ? A.rp1 : (r == 1) ? A.rp2 : A.rp3;
                            }
                            emit(cand >= 0, cand >= 0 ? cand : 0);
                        }
                        for (int i = A.brace_lo; i < A.brace_hi; ++i) emit_height(lane == 0, M.brace_nodes[i]);
                        if (cnt > kMhbList) cnt = -1;
                        __builtin_amdgcn_s_waitcnt(0xc07f);
                        __builtin_amdgcn_wave_barrier();
                        return cnt;
                    };
                    PropApply G;
                    mh_propose_ranges(Mt_, kind, node, G);
                    int c_kind = G.kind, c_hlo = G.hlo, c_hhi = G.hhi, c_hlo2 = G.hlo2, c_hhi2 = G.hhi2, c_rlo = G.rlo, c_rhi = G.rhi, c_pt1 = G.pt1, c_pt2 = G.pt2, c_rp1 = G.rp1, c_rp2 = G.rp2, c_rp3 = G.rp3, c_brace_lo = G.brace_lo, c_brace_hi = G.brace_hi;
                    int cnt = 0;
                    double pcol[4][R];                           // the first four columns, requested ahead of the request
                    for (int pass = 0; pass < 2; ++pass) {
                        cnt = build_list(c_kind, c_hlo, c_hhi, c_hlo2, c_hhi2, c_rlo, c_rhi, c_pt1, c_pt2, c_rp1, c_rp2, c_rp3, c_brace_lo, c_brace_hi, 2 * tag + pass);
                        if (cnt > 0) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int m = (u < cnt) ? u : cnt - 1;
                                const int j = __builtin_amdgcn_readfirstlane(l_j[m]);
                                const double* wc = V.Wc + (size_t)j * NPadC + lane;
#pragma unroll
                                for (int k = 0; k < R; ++k) pcol[u][k] = wc[64 * k];
                            }
                        }
                        if (pass == 1) break;
                        (void)poll(&words->req, tag, 0);
                        const int n_kind = A_lds->kind, n_hlo = A_lds->hlo, n_hhi = A_lds->hhi, n_hlo2 = A_lds->hlo2, n_hhi2 = A_lds->hhi2, n_rlo = A_lds->rlo, n_rhi = A_lds->rhi, n_pt1 = A_lds->pt1, n_pt2 = A_lds->pt2, n_rp1 = A_lds->rp1, n_rp2 = A_lds->rp2, n_rp3 = A_lds->rp3, n_brace_lo = A_lds->brace_lo, n_brace_hi = A_lds->brace_hi;
                        const bool same_ranges = n_kind == c_kind && n_hlo == c_hlo && n_hhi == c_hhi && n_hlo2 == c_hlo2 && n_hhi2 == c_hhi2 && n_rlo == c_rlo && n_rhi == c_rhi && n_pt1 == c_pt1 && n_pt2 == c_pt2 && n_rp1 == c_rp1 && n_rp2 == c_rp2 && n_rp3 == c_rp3 && n_brace_lo == c_brace_lo && n_brace_hi == c_brace_hi;
                        if (same_ranges) break;
                        c_kind = n_kind;
                        c_hlo = n_hlo;
                        c_hhi = n_hhi;
                        c_hlo2 = n_hlo2;
                        c_hhi2 = n_hhi2;
                        c_rlo = n_rlo;
                        c_rhi = n_rhi;
                        c_pt1 = n_pt1;
                        c_pt2 = n_pt2;
                        c_rp1 = n_rp1;
                        c_rp2 = n_rp2;
                        c_rp3 = n_rp3;
                        c_brace_lo = n_brace_lo;
                        c_brace_hi = n_brace_hi;
                    }
                    // the listed slots' new distances from the proposed state and the deltas against the current ones (the arithmetic of
                    // `distances` below)
                    if (cnt > 0) {
                        const double s1 = words->s1;
                        double d0 = 0.0;
                        bool have0 = false;
                        for (int m = lane; m < cnt; m += 64) {
                            const int slot = l_j[m];
                            const int a = ts_node[slot], pa = ts_parent[slot];
                            double x = (Hp_[pa] - Hp_[a]) * Rp_[a];
                            if (slot == 0) x = x + (Hp_[0] - Hp_[rr_]) * Rp_[rr_];
                            x = x * s1;
                            l_dnew[m] = x;
                            l_delta[m] = x - dcur_l[slot];
                            if (slot == 0) {
                                d0 = x;
                                have0 = true;
                            }
                        }
                        const uint64_t m0 = __builtin_amdgcn_ballot_w64(have0);
                        if (lane == 0) words->have0 = (m0 != 0) ? 1 : 0;
                        if (m0 != 0) {
                            const double lj1 = log(1.0 / readlane64(d0, (int)__builtin_ctzll(m0)));      // jacobianRootBranch, :393-410
                            if (lane == 0) words->lj = lj1;
                        }
                        __builtin_amdgcn_s_waitcnt(0xc07f);
                        __builtin_amdgcn_wave_barrier();
                    } else if (lane == 0) {
                        words->have0 = 0;
                    }
                    double zp[R];
#pragma unroll
                    for (int k = 0; k < R; ++k) zp[k] = z_l[64 * k + lane];
                    if (cnt > 0) {                               // the first batch: the columns requested ahead of the request
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const double dl = (u < cnt) ? l_delta[u] : 0.0;     // (past the end: the last column again with weight 0: exact)
#pragma unroll
                            for (int k = 0; k < R; ++k) zp[k] = fma(dl, pcol[u][k], zp[k]);
                        }
                    }
                    for (int m0_ = 4; m0_ < cnt; m0_ += 4) {     // four columns in flight
                        double col[4][R], dl[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int m = (m0_ + u < cnt) ? m0_ + u : cnt - 1;
                            const int j = __builtin_amdgcn_readfirstlane(l_j[m]);
                            dl[u] = (m0_ + u < cnt) ? l_delta[m] : 0.0;
                            const double* wc = V.Wc + (size_t)j * NPadC + lane;
#pragma unroll
                            for (int k = 0; k < R; ++k) col[u][k] = wc[64 * k];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int k = 0; k < R; ++k) zp[k] = fma(dl[u], col[u][k], zp[k]);
                    }
                    double sq = 0.0;
#pragma unroll
                    for (int k = 0; k < R; ++k) sq = fma(zp[k], zp[k], sq);
                    const double q = wave_sum(sq);
                    if (lane == 0) {
                        words->q = q;
                        words->cnt = cnt;
                    }
                    post(&words->resp, tag);
                    const int d = poll(&words->dec, tag, 1);
                    if ((d & 1) && cnt > 0) {
#pragma unroll
                        for (int k = 0; k < R; ++k) z_l[64 * k + lane] = zp[k];
                        for (int m = lane; m < cnt; m += 64) dcur_l[l_j[m]] = l_dnew[m];
                    }
                    post(&words->done, tag);                     // (the chain wave may sweep from / write to distances and z again)
                }
            }
            kind = __builtin_amdgcn_readfirstlane(kind_next);
            node = __builtin_amdgcn_readfirstlane(node_next);
            sp = __builtin_amdgcn_readfirstlane(sp_next);
            p_next = p_next2;
        }
        return;
    }

    // ---- chain waves
    const int64_t B = M.batch;
    const int64_t b_raw = (int64_t)blockIdx.x * CW + wave;
    const bool valid = b_raw < B;                            // a chain beyond the batch works on the last chain's inputs and stores nothing
    const int64_t b = valid ? b_raw : B - 1;
    // tree tables in LDS, shared by the two chains (filled by both chain waves; no workgroup barrier may be used here: the
    // loaders are already at theirs) -- each wave fills the whole table itself, the values are the same
    int32_t* tb_parent = reinterpret_cast<int32_t*>(dyn);
    int32_t* tb_size = tb_parent + nn;
    int32_t* tb_first = tb_size + nn;
    int32_t* tb_nch = tb_first + nn;
    int32_t* tb_second = tb_nch + nn;
    for (int v = lane; v < nn; v += 64) {
        tb_parent[v] = M.parent[v];
        tb_size[v] = M.size[v];
        tb_first[v] = P.first_child[v];
        tb_nch[v] = P.n_children[v];
        tb_second[v] = P.second_child[v];
    }
    MhDev Ml = M;
    Ml.parent = tb_parent;
    Ml.size = tb_size;
    PriorDev Pl = P;
    if (mhb_node_tables_bytes(nn, NP, R, P.n_cal, P.n_con) > 0)      // (each chain wave copies them: the same values; its own copies are what it reads)
        prior_stage_node_tables(Pl, P, dyn + mhb_lds_bytes(nn, NP, R) / sizeof(double), lane, 64);
    Pl.parent = tb_parent;
    Pl.first_child = tb_first;
    Pl.n_children = tb_nch;
    Pl.second_child = tb_second;
    double* Hc = dyn + mhb_table_doubles(nn) + (size_t)wave * mhb_chain_doubles(nn, NP);
    double* Rc = Hc + nn;
    double* Hp = Rc + nn;
    double* Rp = Hp + nn;
    const double* tune = M.tune + b * NP;                    // (constant during a launch: mcd_mh_tune is a call of its own)
    int32_t* acc = M.acc + b * NP;
    int32_t* tried = M.tried + b * NP;
    double* tbd_cur = Rp + nn;                               // summand of node v in the birth-death block, current state
    double* tbd_prop = tbd_cur + nn;                         // ... proposed state, valid where stamp_bd[v] = this step
    double* tcl_cur = tbd_prop + nn;                         // the same for the clock block
    double* tcl_prop = tcl_cur + nn;
    int32_t* stamp_bd = reinterpret_cast<int32_t*>(tcl_prop + nn);
    int32_t* stamp_cl = stamp_bd + nn;
    if constexpr (HLP) {
        // (the likelihood waves use these after the first ring barrier -- a launch with columns of L^-1 begins with a sweep of the
        // current state -- and both chain waves write the same shared values)
        for (int v = lane; v < nn; v += 64) ts_of[v] = -1;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < NPadC; j += 64) {
            const int a = T.slot_node[j];
            ts_node[j] = (int16_t)a;
            ts_parent[j] = (int16_t)T.slot_parent[j];
            if (a >= 0) ts_of[a] = (int16_t)j;
            mark[j] = 0;
        }
        if (lane == 0) {
            ts_of[T.root_right] = 0;                         // the root's two daughters share slot 0 (sumFirstTwo)
            words->req = 0;
            words->resp = 0;
            words->dec = 0;
            words->done = 0;
            words->cnt = 0;
            words->have0 = 0;
        }
    }
    for (int w = lane; w < nn; w += 64) {
        stamp_bd[w] = 0;
        stamp_cl[w] = 0;
        Hc[w] = M.H[b * M.ld + w];
        Rc[w] = M.R[b * M.ld + w];
        Hp[w] = Hc[w];                                       // invariant between steps: proposed arrays = current arrays
        Rp[w] = Rc[w];
    }
    double sc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) sc[i] = M.sc[i * B + b];
    double ll = M.post[B + b], lj = M.post[2 * B + b];
    __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): this wave's LDS writes have landed (one wave: in order)
    __builtin_amdgcn_wave_barrier();
    // the three blocks of the ln prior of the current state; a step re-evaluates only the blocks whose inputs moved
    double c0 = prior_nodes_wave(Pl, lane, sc[2], Hc);
    // The birth-death and the clock block are sums of one summand per node (prior_device.hpp: prior_bd_term, prior_clock_term).  The
    // summands of the current state are kept; a proposal that writes a few nodes re-evaluates the summands of those nodes (and,
    // for the birth-death block, of their daughters) in ONE pass with the nodes compacted onto the first lanes, and the sum is
    // taken over kept and new summands in the order of the full evaluation: the same function values added in the same order, the
    // same bits as prior_bd_wave / prior_clock_wave -- at one summand's latency instead of four pipelined ones.
    auto bd_full = [&](double la_, double mu_, const double* Hx, double* store) -> double {
        const bool near = prior_bd_near(la_, mu_);
        double bd = 0.0;
        for (int v = 1 + lane; v < nn; v += 64) {
            const double t = prior_bd_term(Pl, v, near, la_, mu_, Hx);
            store[v] = t;
            bd += t;
        }
        return prior_bd_finish(pr_wave_sum(bd), la_, mu_);
    };
    auto clock_full = [&](double rm_, double va_, const double* Hx, const double* Rx, ClockCache& c, double* store) -> double {
        if (c.va != va_) prior_clock_scalars(va_, c);
        double cl = 0.0;
        for (int v = 1 + lane; v < nn; v += 64) {
            const double t = prior_clock_term(Pl, v, va_, c.lg_k, c.log_t, Hx, Rx);
            store[v] = t;
            cl += t;
        }
        return prior_clock_finish(Pl, pr_wave_sum(cl), rm_, va_, c.hyper);
    };
    double c1 = bd_full(sc[0], sc[1], Hc, tbd_cur);
    ClockCache cc{__builtin_nan(""), 0.0, 0.0, 0.0};
    double c2 = clock_full(sc[3], sc[4], Hc, Rc, cc, tcl_cur);
    double lp = c0 + c1 + c2;
    // rows 64 k + lane of the solve: mean, 1 / L_ii, the node whose branch feeds the distance slot and that node's parent
    double mu_r[R], iv_r[R];
    int sl_a[R], sl_pa[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int row = 64 * k + lane;
        mu_r[k] = V.mu[row];
        iv_r[k] = V.invdiag[row];
        sl_a[k] = T.slot_node[row];                          // -1 for padded rows
        sl_pa[k] = T.slot_parent[row];                       // 0 for padded rows
    }
    const int rr = T.root_right;
    // likelihoodFunctionWrapper: distances = (tH * rMu) * sumFirstTwo (times * rates)      (app/Probability.hs:195-207), the
    // arithmetic of load_tree (mvn_device.hpp); v[k] = distance of row 64 k + lane (0 in the padding), returns the root slot's
    auto distances = [&](const double* Hx, const double* Rx, double s_, double (&v)[R]) -> double {
        double dist0 = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int rw = 64 * k + lane;
            double x = 0.0;
            if (sl_a[k] >= 0) {
                x = (Hx[sl_pa[k]] - Hx[sl_a[k]]) * Rx[sl_a[k]];
                if (rw == 0) x = x + (Hx[0] - Hx[rr]) * Rx[rr];
                x = x * s_;
            }
            if (k == 0) dist0 = x;
            v[k] = x;
        }
        return dist0;
    };
    // z = L^-1 (v - mu) by the forward sweep against the ring (the loaders stream alongside); returns |z|^2
    auto sweep = [&](const double (&v)[R], double (&z)[R]) -> double {
        double d[R][1];
#pragma unroll
        for (int k = 0; k < R; ++k) d[k][0] = (v[k] - mu_r[k]) * iv_r[k];
        lds_barrier();
        fwd_compute<R, 1, 0>(d, ring, lane, ncols MCD_ACC_ARGS);
        double sq = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            z[k] = d[k][0];
            sq = fma(d[k][0], d[k][0], sq);
        }
        return wave_sum(sq);
    };
    const bool inc = V.Wc != nullptr;                        // incremental evaluation of sparse proposals available
    const int NPad = 64 * R;
    double dcur[HLP ? 1 : R], zc[HLP ? 1 : R];               // distances and z = L^-1 (d - mu) of the CURRENT state (HLP: in LDS)
    if constexpr (HLP) {
        double d0_[R];
        (void)distances(Hc, Rc, sc[2] * sc[3], d0_);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            dcur_l[64 * k + lane] = d0_[k];
            z_l[64 * k + lane] = 0.0;
        }
    } else {
        (void)distances(Hc, Rc, sc[2] * sc[3], dcur);
#pragma unroll
        for (int k = 0; k < R; ++k) zc[k] = 0.0;
    }
    int last_sparse = 0;                                     // (HLP) the last step + 1 the likelihood wave was asked about
    auto wait_done = [&]() {                                 // ... it has committed that step: distances and z are the chain wave's to touch
        if constexpr (HLP) (void)poll(&words->done, last_sparse, 0);
    };
    const double beta = M.beta[b];
#ifdef MCD_MHB_STAMP
    // diagnostic build (make stamp_mhbig): s_memtime ticks per phase, summed over the run, in the first rows of trace_alpha
    uint64_t tk[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // ... 6: column update of sparse steps (4: sweep of dense steps), 7: number of sparse steps, 8: the draws of 64 steps, 9: distances alone
#define MHB_TICK(i)                                       \
    {                                                     \
        const uint64_t now_ = __builtin_readcyclecounter(); \
        tk[i] += now_ - t_last;                           \
        t_last = now_;                                    \
    }
    uint64_t t_last = __builtin_readcyclecounter();
#else
#define MHB_TICK(i)
#endif
    // The nodes a proposal writes (PropApply: up to three pre-order ranges, five single nodes, the braced nodes with their
    // daughters): f(w) for each of them, lanes in parallel (a node may come twice).  Between steps the proposed arrays equal the
    // current ones, so a step applies, commits or takes back its proposal on these nodes only -- most proposals write one to
    // three nodes of the 257, and the copies over all nodes were a fifth of the proposal and most of the accept phase.
    auto for_write_set = [&](const PropApply& A, auto&& f) {
        for (int w = A.hlo + lane; w < A.hhi; w += 64) f(w);
        for (int w = A.hlo2 + lane; w < A.hhi2; w += 64) f(w);
        if (A.rlo != A.hlo || A.rhi != A.hhi)
            for (int w = A.rlo + lane; w < A.rhi; w += 64) f(w);
        const int pt = (lane == 0) ? A.pt1 : (lane == 1) ? A.pt2 : (lane == 2) ? A.rp1 : (lane == 3) ? A.rp2 : (lane == 4) ? A.rp3 : -1;
        if (pt >= 0) f(pt);
        for (int i = A.brace_lo; i < A.brace_hi; ++i) {
            const int x = M.brace_nodes[i];
            const int w = (lane == 0) ? x : (lane == 1 && tb_nch[x] > 0) ? tb_first[x] : (lane == 2 && tb_nch[x] > 1) ? tb_second[x] : -1;
            if (w >= 0) f(w);
        }
    };
    int p = sched[0];
    const int vz = mh_vzero();                               // (mh_device.hpp: what travels ahead is loaded by vector loads)
    int p_next = sched[(n_steps > 1 ? 1 : 0) + vz];          // (the schedule's entries two steps ahead: a row's loads need its index)
    PropRow row = mh_load_row(M, p);
    double t_cur = tune[p];
    int row_sparse = inc ? M.sparse[p] : 0;
    StepDraws pre{1.0, 0.0, 0.0, 0.5, 0.5};                  // lane l: the state-independent draws of step (gs & ~63) + l
    for (int64_t gs = 0; gs < n_steps; ++gs) {
        const int p_next2 = sched[((gs + 2 < n_steps) ? gs + 2 : gs) + vz];
        const PropRow row_next = mh_load_row_ahead(M, p_next);   // the next step's row travels while this step computes
        const double t_next = tune[p_next];                  // ... and its tuning parameter (global memory: a load at the point of use stalled the proposal)
        const int sparse_next = inc ? M.sparse[p_next] : 0;
        if ((gs & 63) == 0) {
            // 64 consecutive steps at once, one step per lane: what can be drawn knowing only the proposal row and its tuning
            // parameter (as k_mh_chain.hip, and as k_mh_draws does for the two-launch path)
            MHB_TICK(0)
            const int64_t mine = gs + lane;
            if (mine < n_steps) {
                const int pl = sched[mine];
                pre = mh_step_draws(mh_load_row(M, pl), tune[pl], mh_rng(seed, M.chain0 + b, step0 + (uint64_t)mine));
            }
            MHB_TICK(8)
        }
        const int sl = (int)(gs & 63);
        const StepDraws dr{mh_readlane64(pre.u, sl), mh_readlane64(pre.lnq, sl), mh_readlane64(pre.logu, sl), mh_readlane64(pre.U, sl),
                           mh_readlane64(pre.Uacc, sl)};
        double sc1[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) sc1[i] = sc[i];
        MHB_TICK(0)
        PropApply A;
        const double lnqj = mh_propose_params(Ml, row, t_cur, dr, lane, sc1, Hc, Rc, A);
        for_write_set(A, [&](int w) {
            double h, r;
            mh_propose_node(Ml, A, w, Hc, Rc, h, r);
            Hp[w] = h;
            Rp[w] = r;
        });
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): the writes above have landed before any lane reads them
        __builtin_amdgcn_wave_barrier();
        MHB_TICK(1)
        // distances of the proposed state; for a sparse row: which of them moved, and the first columns of L^-1 on their way
        const bool moves = mhb_moves_likelihood(row.kind, row.node);      // (a function of the row alone: the loaders decide alike)
        const bool sparse_step = moves && row_sparse != 0;
        const int tag = (int)gs + 1;
        double vp[R], dl[HLP ? 1 : R];
        double lj1 = lj;
        uint64_t mk[HLP ? 1 : R];
        int cj[MhbCols<R>::N];
        double cd[MhbCols<R>::N], col[HLP ? 1 : MhbCols<R>::N][HLP ? 1 : R];
        int ccnt = 0;
        if constexpr (!HLP) {
#pragma unroll
            for (int k = 0; k < R; ++k) mk[k] = 0;
        }
        auto fetch_cols = [&]() {                             // up to MhbCols<R>::N moved distances, in row order: their columns requested
            ccnt = 0;
            if constexpr (!HLP) {
#pragma unroll
            for (int c = 0; c < MhbCols<R>::N; ++c) {
                int j = (c > 0) ? cj[c - 1] : 0;
                double d_ = 0.0;                              // no more: the previous column again with weight 0 (exact)
                bool got = false;
#pragma unroll
                for (int k = 0; k < R; ++k) {                 // (wave-uniform; no early exit, so that the arrays stay registers)
                    const bool take = !got && mk[k] != 0;
                    if (take) {
                        const int l = (int)__builtin_ctzll(mk[k]);
                        mk[k] &= mk[k] - 1;
                        j = 64 * k + l;
                        d_ = readlane64(dl[k], l);
                    }
                    got = got || take;
                }
                ccnt += got ? 1 : 0;
                cj[c] = j;
                cd[c] = d_;
                const double* wc = V.Wc + (size_t)j * NPad + lane;
#pragma unroll
                for (int k = 0; k < R; ++k) col[c][k] = wc[64 * k];
            }
            }
        };
        // (HLP) a sparse step: the likelihood wave takes it from here -- the transform (which nodes are written), Hp / Rp, tH * rMu.
        // On a step that begins with a recomputation of z (every 256 steps) the request waits until that sweep has written z: the
        // likelihood wave, which streams for the sweep first, must not read the old one (and the sweep waits for its last commit).
        const bool refresh_first = inc && (gs & (kMhbRefresh - 1)) == 0;
        auto request = [&]() {
            if (lane == 0) {
                *A_lds = A;
                words->s1 = sc1[2] * sc1[3];
            }
            post(&words->req, tag);
            last_sparse = tag;
        };
        if (HLP && sparse_step) {
            if (!refresh_first) request();
        } else if (moves) {
            const double dist0 = distances(Hp, Rp, sc1[2] * sc1[3], vp);
            lj1 = log(1.0 / readlane64(dist0, 0));          // jacobianRootBranch, :393-410
            MHB_TICK(9)
            if constexpr (!HLP) {
            if (sparse_step) {
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    dl[k] = vp[k] - dcur[k];
                    mk[k] = __builtin_amdgcn_ballot_w64(vp[k] != dcur[k]);   // NaN != x: counted, and the NaN then reaches q
                }
                fetch_cols();
            }
            }
        }
        MHB_TICK(3)
        // which arrays the proposal writes (a superset of "changed": a block re-evaluated on unchanged inputs returns the same bits)
        const bool dH = A.hhi > A.hlo || A.hhi2 > A.hlo2 || A.pt1 >= 0 || A.pt2 >= 0 || A.brace_hi > A.brace_lo;
        const bool dR = A.rhi > A.rlo || A.rp1 >= 0 || A.rp2 >= 0 || A.rp3 >= 0 || (A.brace_hi > A.brace_lo && A.kind == MCD_PROP_SLIDE_BRACE_CONTRA);
        ClockCache ccp = cc;                                 // refreshed only if the proposal moved rVar
        const double c0p = (dH || sc1[2] != sc[2]) ? prior_nodes_wave(Pl, lane, sc1[2], Hp) : c0;
        const int st = (int)(gs + 1);                        // this step's mark on proposed summands
        const int nbr = A.brace_hi - A.brace_lo;
        // candidate l of the birth-death block: the nodes whose height the proposal writes, with their daughters (a range is a sub
        // tree without its root: closed under "daughter of"); -1 = none
        const int len1 = A.hhi > A.hlo ? A.hhi - A.hlo : 0, len2 = A.hhi2 > A.hlo2 ? A.hhi2 - A.hlo2 : 0;
        auto cand_bd = [&](int l) -> int {
            if (l < len1) return A.hlo + l;
            l -= len1;
            if (l < len2) return A.hlo2 + l;
            l -= len2;
            const int g = l / 3, r = l - 3 * g;
            int base = -1;
            if (g == 0) base = A.pt1; else if (g == 1) base = A.pt2; else if (g - 2 < nbr) base = M.brace_nodes[A.brace_lo + g - 2];
            if (base < 0) return -1;
            if (r == 0) return base;
            return (tb_nch[base] >= r) ? (r == 1 ? tb_first[base] : tb_second[base]) : -1;
        };
        const int cnt_bd = len1 + len2 + 3 * (2 + nbr);
        // ... of the clock block (uncorrelated models: a summand depends on its node's rate only): the nodes whose rate is written
        const int lenr = A.rhi > A.rlo ? A.rhi - A.rlo : 0;
        const int nbr_r = (A.kind == MCD_PROP_SLIDE_BRACE_CONTRA) ? nbr : 0;
        auto cand_cl = [&](int l) -> int {
            if (l < lenr) return A.rlo + l;
            l -= lenr;
            if (l < 3) return l == 0 ? A.rp1 : l == 1 ? A.rp2 : A.rp3;
            l -= 3;
            const int g = l / 3, r = l - 3 * g;
            if (g >= nbr_r) return -1;
            const int base = M.brace_nodes[A.brace_lo + g];
            if (r == 0) return base;
            return (tb_nch[base] >= r) ? (r == 1 ? tb_first[base] : tb_second[base]) : -1;
        };
        const int cnt_cl = lenr + 3 + 3 * nbr_r;
        const bool bd_scalars = sc1[0] != sc[0] || sc1[1] != sc[1];
        const bool need_bd = dH || bd_scalars;
        const bool few_bd = need_bd && !bd_scalars && cnt_bd <= 64 && !prior_bd_near(sc1[0], sc1[1]);
        double c1p = c1;
        // (HLP, as k_mh_segment.hip: a proposal's few summands are written IN PLACE, the old values wait in registers for the decision --
        // the sum then reads one array instead of three, an accepted proposal has nothing to copy.  A node may come twice: every lane reads
        // the old value before any lane writes, LDS keeps a wave's order)
        double old_bd = 0.0, old_cl = 0.0;
        const int v_bd = few_bd ? cand_bd(lane) : -1;
        const bool mine_bd = few_bd && lane < cnt_bd && v_bd >= 1;
        if (few_bd) {
            if constexpr (HLP) {
                if (mine_bd) old_bd = tbd_cur[v_bd];
                const double t = mine_bd ? prior_bd_term(Pl, v_bd, false, sc1[0], sc1[1], Hp) : 0.0;
                if (mine_bd) tbd_cur[v_bd] = t;
            } else if (mine_bd) {
                tbd_prop[v_bd] = prior_bd_term(Pl, v_bd, false, sc1[0], sc1[1], Hp);
                stamp_bd[v_bd] = st;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            double bd = 0.0;
            if constexpr (HLP) {
                for (int w = 1 + lane; w < nn; w += 64) bd += tbd_cur[w];
            } else {
                for (int w = 1 + lane; w < nn; w += 64) bd += (stamp_bd[w] == st) ? tbd_prop[w] : tbd_cur[w];
            }
            c1p = prior_bd_finish(pr_wave_sum(bd), sc1[0], sc1[1]);
        } else if (need_bd) {
            c1p = bd_full(sc1[0], sc1[1], Hp, tbd_prop);
        }
        const bool cl_heights = dH && P.clock_model >= 2;    // white noise / autocorrelated: the summands also hold branch durations
        const bool need_cl = dR || sc1[3] != sc[3] || sc1[4] != sc[4] || cl_heights;
        const bool few_cl = need_cl && sc1[4] == sc[4] && P.clock_model < 2 && cnt_cl <= 64;
        double c2p = c2;
        const int v_cl = few_cl ? cand_cl(lane) : -1;
        const bool mine_cl = few_cl && lane < cnt_cl && v_cl >= 1;
        if (few_cl) {
            if constexpr (HLP) {
                if (mine_cl) old_cl = tcl_cur[v_cl];
                const double t = mine_cl ? prior_clock_term(Pl, v_cl, sc1[4], cc.lg_k, cc.log_t, Hp, Rp) : 0.0;
                if (mine_cl) tcl_cur[v_cl] = t;
            } else if (mine_cl) {
                tcl_prop[v_cl] = prior_clock_term(Pl, v_cl, sc1[4], cc.lg_k, cc.log_t, Hp, Rp);
                stamp_cl[v_cl] = st;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            double cl = 0.0;
            if constexpr (HLP) {
                for (int w = 1 + lane; w < nn; w += 64) cl += tcl_cur[w];
            } else {
                for (int w = 1 + lane; w < nn; w += 64) cl += (stamp_cl[w] == st) ? tcl_prop[w] : tcl_cur[w];
            }
            c2p = prior_clock_finish(Pl, pr_wave_sum(cl), sc1[3], sc1[4], cc.hyper);
        } else if (need_cl) {
            c2p = clock_full(sc1[3], sc1[4], Hp, Rp, ccp, tcl_prop);
        }
        const double lp1 = c0p + c1p + c2p;
        MHB_TICK(2)
        double ll1 = ll;
        double zp[R];
        // The sweeps of this step, ONE call site (the sweep is a long unrolled body): first z of the CURRENT state when it is due
        // (start of the launch, then every 256 steps: the loaders stream for it under the same condition), then the step's own if its
        // proposal is dense.
        const bool refresh_now = inc && (gs & (kMhbRefresh - 1)) == 0;
        const bool dense_step = moves && !sparse_step;
        for (int pass = 0; pass < 2; ++pass) {
            const bool rf = pass == 0 && refresh_now;
            if (!rf && !(pass == 1 && dense_step)) continue;
            double vin[R], zout[R];
            if constexpr (HLP) {
                if (rf) wait_done();                          // (the likelihood wave's last commit is in LDS)
#pragma unroll
                for (int k = 0; k < R; ++k) vin[k] = rf ? dcur_l[64 * k + lane] : vp[k];
            } else {
#pragma unroll
                for (int k = 0; k < R; ++k) vin[k] = rf ? dcur[k] : vp[k];
            }
            const double q = sweep(vin, zout);
            if (rf) {
                if constexpr (HLP) {
#pragma unroll
                    for (int k = 0; k < R; ++k) z_l[64 * k + lane] = zout[k];
                } else {
#pragma unroll
                    for (int k = 0; k < R; ++k) zc[k] = zout[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < R; ++k) zp[k] = zout[k];
                ll1 = V.c + (-0.5) * (V.logdet + q);         // :169 (finish_ll)
            }
        }
        if constexpr (!HLP) {
        if (!dense_step) {
#pragma unroll
            for (int k = 0; k < R; ++k) zp[k] = zc[k];
        }
        }
        if (HLP && sparse_step) {
            // the likelihood wave's answer
            if (refresh_first) request();
            (void)poll(&words->resp, tag, 0);
            const double q = words->q;
            const int cnt = words->cnt;
            if (words->have0) lj1 = words->lj;
            if (cnt < 0) ll = __builtin_nan("");             // (more moved distances than the list holds: cannot happen for a row marked
                                                             // sparse; if it does the chain says so -- its ln likelihood is NaN from here on)
            ll1 = (cnt < 0) ? __builtin_nan("") : V.c + (-0.5) * (V.logdet + q);     // :169 (finish_ll)
        } else if (sparse_step) {
            // z' = z + sum_j delta_j W[:, j] over the moved distances, in row order; q' = |z'|^2
            while (true) {
#pragma unroll
                for (int c = 0; c < MhbCols<R>::N; ++c)
#pragma unroll
                    for (int k = 0; k < R; ++k) zp[k] = fma(cd[c], col[c][k], zp[k]);
                if (ccnt < MhbCols<R>::N) break;
                fetch_cols();                                 // (more than MhbCols<R>::N moved: another round trip)
                if (ccnt == 0) break;
            }
            double sq = 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) sq = fma(zp[k], zp[k], sq);
            const double q = wave_sum(sq);
            ll1 = V.c + (-0.5) * (V.logdet + q);             // :169 (finish_ll)
        }
#ifdef MCD_MHB_STAMP
        if (sparse_step) { MHB_TICK(6) tk[7] += 1; } else { MHB_TICK(4) }
#endif
        double la = beta * ((lp1 + ll1) - (lp + ll)) + lnqj;           // heated chains of MC3: posterior^beta; beta = 1 is exact
        if (row.jac_root) la += (double)row.jac_root * (lj1 - lj);
        const bool ok = (la >= 0) || (dr.Uacc < exp(la));
        if (HLP && sparse_step) post(&words->dec, 2 * tag + (ok ? 1 : 0));
        if (ok) {
            for_write_set(A, [&](int w) {
                Hc[w] = Hp[w];
                Rc[w] = Rp[w];
            });
#pragma unroll
            for (int i = 0; i < 5; ++i) sc[i] = sc1[i];
            if (few_bd) {
                if constexpr (!HLP) {
                    if (mine_bd) tbd_cur[v_bd] = tbd_prop[v_bd];
                }
            } else if (need_bd) {
                for (int w = 1 + lane; w < nn; w += 64) tbd_cur[w] = tbd_prop[w];
            }
            if (few_cl) {
                if constexpr (!HLP) {
                    if (mine_cl) tcl_cur[v_cl] = tcl_prop[v_cl];
                }
            } else if (need_cl) {
                for (int w = 1 + lane; w < nn; w += 64) tcl_cur[w] = tcl_prop[w];
            }
            c0 = c0p;
            c1 = c1p;
            c2 = c2p;
            cc = ccp;
            lp = lp1;
            ll = ll1;
            lj = lj1;
            if constexpr (HLP) {
                if (dense_step) {                             // (a sparse proposal's distances and z': the likelihood wave's to commit)
                    wait_done();
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        z_l[64 * k + lane] = zp[k];
                        dcur_l[64 * k + lane] = vp[k];
                    }
                }
            } else if (moves) {
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    zc[k] = zp[k];
                    dcur[k] = vp[k];
                }
            }
        }
        if (!ok) {
            for_write_set(A, [&](int w) {
                Hp[w] = Hc[w];
                Rp[w] = Rc[w];
            });
            if constexpr (HLP) {                              // the overwritten summands back
                if (mine_bd) tbd_cur[v_bd] = old_bd;
                if (mine_cl) tcl_cur[v_cl] = old_cl;
            }
        }
        if (lane == 0) {
            if (valid) {
                atomicAdd(&tried[p], 1);                     // (global memory, no value returned: nothing waits for it -- a load-add-store
                if (ok) atomicAdd(&acc[p], 1);               // made the wave wait for the load: 257 nodes 7.33 -> 7.16 us per lock step)
                if (trace_alpha) trace_alpha[gs * B + b] = la;
                if (trace_accept) trace_accept[gs * B + b] = ok ? 1 : 0;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (accumulate && valid && (gs + 1) % S == 0) {      // once per iteration of the cycle: straight into the running sums
            for (int w = lane; w < nn; w += 64) {
                const double a = sc[2] * Hc[w];
                M.age_sum[b * nn + w] += a;
                M.age_sq[b * nn + w] += a * a;
            }
        }
        p = __builtin_amdgcn_readfirstlane(p_next);
        p_next = p_next2;
        row = mh_row_scalar(row_next);
        t_cur = t_next;
        row_sparse = __builtin_amdgcn_readfirstlane(sparse_next);
        MHB_TICK(5)
    }
#ifdef MCD_MHB_STAMP
    if (trace_alpha && lane == 0 && valid)
        for (int i = 0; i < 10; ++i) trace_alpha[(int64_t)i * B + b] = (double)tk[i];   // ticks: loop head, propose, prior, distances + column requests, sweep or column update, accept
#endif
    if (!valid) return;
    for (int w = lane; w < nn; w += 64) {
        M.H[b * M.ld + w] = Hc[w];
        M.R[b * M.ld + w] = Rc[w];
    }
    if (lane < 5) {
        double mine = sc[0];
#pragma unroll
        for (int i = 1; i < 5; ++i)
            if (lane == i) mine = sc[i];
        M.sc[lane * B + b] = mine;
    }
    if (lane == 0) {
        M.post[b] = lp;
        M.post[B + b] = ll;
        M.post[2 * B + b] = lj;
        M.pcomp[b * 3 + 0] = c0;                             // (a later run on the two-launch path continues from these)
        M.pcomp[b * 3 + 1] = c1;
        M.pcomp[b * 3 + 2] = c2;
    }
}



// trees of 65 .. 514 nodes whose factor the sweep holds in 2 .. 8 register blocks (N <= 512), a batch of at most two rounds of
// workgroups (one workgroup per CU: 512 chains per round), state + tables + the 64 KiB ring within a CU's LDS
bool mh_chain_big_available(const MhDev& M, const MvnDev& V)
{
    // (R <= 4: up to 258 nodes.  Round 3 also built R = 6 and 8 -- 514 nodes -- which the segment kernel has superseded from 259 nodes and which
    // did not fit the register file: 204 / 420 bytes of scratch per lane; removed in round 4.  1024 chains: two rounds of workgroups, still
    // ahead of two launches per step)
    if (V.R < 1 || V.R > 4 || M.n_nodes > 64 * V.R + 2 || M.batch > 1024) return false;
    return mhb_lds_bytes(M.n_nodes, M.n_prop, V.R) + 64 * 1024 <= 160 * 1024;
}

template <int R>
static hipError_t launch_big_R(const MhDev& M, const MvnDev& V, const TreeDev& T, const PriorDev& P, const int32_t* sched, int64_t n_steps,
                               int32_t S, int accumulate, uint64_t step0, uint64_t seed, double* trace_alpha, int8_t* trace_accept, hipStream_t st)
{
    const size_t dynb = mhb_lds_bytes(M.n_nodes, M.n_prop, R) + mhb_node_tables_bytes(M.n_nodes, M.n_prop, R, P.n_cal, P.n_con);
    static std::atomic<unsigned long long> allowed{0};       // more than 64 KiB of LDS in total has to be allowed once per device
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!((allowed.load(std::memory_order_acquire) >> dev) & 1ull)) {
        if (hipError_t e = hipFuncSetAttribute((const void*)k_mh_chain_big<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) return e;
        allowed.fetch_or(1ull << dev, std::memory_order_release);
    }
    note_dynamic_lds(dynb + 64 * 1024);                      // (+ the sweep's static LDS ring)
    hipLaunchKernelGGL(k_mh_chain_big<R>, dim3((unsigned)((M.batch + 1) / 2)), dim3(256), dynb, st, M, V, T, P, sched, n_steps, S, accumulate,
                       step0, seed, trace_alpha, trace_accept);
    return hipGetLastError();
}

hipError_t launch_mh_chain_big(const MhDev& M, const MvnDev& V, const TreeDev& T, const PriorDev& P, const int32_t* sched, int64_t n_steps,
                               int32_t S, int accumulate, uint64_t step0, uint64_t seed, double* trace_alpha, int8_t* trace_accept,
                               hipStream_t st)
{
    if (n_steps <= 0) return hipSuccess;
    if (!mh_chain_big_available(M, V)) return hipErrorInvalidValue;
    if (opt_is(OPT_MH_INCREMENTAL, 0)) {                     // (mcd_set_option "MCD_MH_INCREMENTAL" = 0: every proposal through the full sweep; tests, timing)
        MvnDev V0 = V;
        V0.Wc = nullptr;
        switch (V.R) {
        case 1: return launch_big_R<1>(M, V0, T, P, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, st);
        case 2: return launch_big_R<2>(M, V0, T, P, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, st);
        case 3: return launch_big_R<3>(M, V0, T, P, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, st);
        case 4: return launch_big_R<4>(M, V0, T, P, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, st);
        default: return hipErrorInvalidValue;
        }
    }
    switch (V.R) {
    case 1: return launch_big_R<1>(M, V, T, P, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, st);       // (65 and 66 nodes: one more than the small-tree kernel holds)
    case 2: return launch_big_R<2>(M, V, T, P, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, st);
    case 3: return launch_big_R<3>(M, V, T, P, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, st);
    case 4: return launch_big_R<4>(M, V, T, P, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mcd
