// k_mh_segment.hip -- Metropolis-Hastings-Green on trees of 259 .. 1026 nodes at a sampler's batch (gfx950): a SEGMENT of a schedule
// in one launch.  SURVEY.md 8(f) row f2.
//
// On the larger of these trees the factor of Sigma (4.2 MB at N = 1023) cannot be streamed per chain and step as k_mh_chain_big.hip
// does up to 514 nodes (and from 259 nodes that stream, 1 MB per workgroup and dense step at 513 nodes, costs more than it saves:
// 513 nodes x 512 chains 16.0 us per lock step there, 9.6 here): a proposal that moves many branch distances takes the two-launch path (k_mh.hip proposes, the row-split kernel
// evaluates, k_mh.hip accepts).  But most proposals of the cycle move a few distances, and that path costs them the same two
// trips of every chain's whole state through memory: 61 KB in and as much out per chain and step, 60 MB per lock step at 512
// chains -- the memory system, not the arithmetic, set its 26 us.  mh_capi.cpp therefore cuts the schedule at the dense proposals
// and runs every stretch between two of them (and between two recomputations of z, every 256 steps) as ONE launch of this kernel.
//
// A workgroup owns TWO chains for the whole segment, each with a CHAIN wave and a LIKELIHOOD wave:
//   chain wave       state (heights, rates, scalars) and the kept per-node summands of the ln prior -- all in LDS from the first step
//                    to the last; per step: propose (mh_device.hpp) on the nodes the proposal writes, post the transform -> ln prior:
//                    the summands of the written nodes in place, the old values kept in registers (prior_device.hpp; sums in the
//                    order of the full evaluation: the same bits as k_mh.hip) -> wait for |z'|^2 -> accept / reject -> commit or
//                    take back on the written nodes only
//   likelihood wave  the current distances (LDS) and z = L^-1 (d - mu) of the current state (registers, R per lane); per step, while
//                    the chain wave evaluates the ln prior: the distance slots the written nodes feed, each once (an LDS exchange per
//                    candidate slot), their new distances and deltas as a list -> z' = z + sum_j delta_j W[:, j] over the list, four
//                    columns of W = L^-1 (MvnDev::Wc, 8 KiB each at N = 1023, an L2 miss each) in flight -> |z'|^2 (and ln
//                    jacobianRootBranch when slot 0 moved) back through LDS; keeps z' and the new distances when the chain wave says so
//                    (measured: 3 / 4 / 6 / 8 columns in flight 11.27 / 11.21 / 11.62 / 12.03 us per lock step at 1025 nodes -- the
//                    padding of a batch is loaded too, and most proposals move one to three distances)
// The two waves of a chain talk through a few LDS words (request, reply, decision; count, |z'|^2, the transform): no workgroup barrier after the tables
// are in LDS, nothing leaves the CU.  A chain's state, distances, z and summands come from where the two-launch path keeps them (MhDev, MhInc::X0 / zcur,
// MhDev::psum) and go back there at the end: that path continues from them.
//
// Parity: proposals, priors, decisions as k_mh.hip (same functions, same numbers, same order of summation); the ln likelihood of
// a proposal agrees with a full evaluation to rounding, as in every incremental path (k_mh_inc.hip), z recomputed by a full
// product every 256 steps between two segments.  tests/test_gpu_mh.py::test_incremental_likelihood_on_large_trees runs the same
// chains with and without segments and without any incremental evaluation: identical decisions, states and ln priors.
//
// Reference: the loop this replaces is `mhg`'s iteration of `mcmc` [external] driven from app/Main.hs:460-479 with the cycle of
// app/Definitions.hs:256-278; likelihood app/Probability.hs:166-173, 195-207; jacobianRootBranch :393-410.
#include "mvn_device.hpp"
#include "mh_segment_device.hpp"
#include "options.h"

#include <atomic>

namespace mcd {

// HELP: two more waves per chain evaluate the birth-death and the clock block of the ln prior beside the chain wave (mh_segment_device.hpp)
template <int R, bool HELP>
__global__ __launch_bounds__(HELP ? 512 : 256, 1) void k_mh_segment(MhDev M, MvnDev V, TreeDev T, PriorDev P, MhInc I, const int32_t* __restrict__ sched,
                                                    int64_t n_steps, int32_t S, int accumulate, uint64_t step0, uint64_t seed,
                                                    double* __restrict__ trace_alpha, int8_t* __restrict__ trace_accept, int64_t gs_base,
                                                    int summands_kept, MhSegPending Q)
{
    extern __shared__ double dyn[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cs = wave & 1;                                 // which of the workgroup's two chains
    const int role = wave >> 1;                              // 0 the chain wave, 1 the likelihood wave, 2 / 3 the prior waves (HELP)
    constexpr int NT = HELP ? 512 : 256;
    const int nn = M.n_nodes;
    const int NPad = 64 * R;
    const int64_t B = M.batch;
    const int64_t b_raw = (int64_t)blockIdx.x * 2 + cs;
    const bool valid = b_raw < B;                            // a chain beyond the batch works on the last chain's inputs and stores nothing
    const int64_t b = valid ? b_raw : B - 1;
    // ---- LDS
    int32_t* tb_parent = reinterpret_cast<int32_t*>(dyn);
    int32_t* tb_size = tb_parent + nn;
    int32_t* tb_first = tb_size + nn;
    int32_t* tb_nch = tb_first + nn;
    int32_t* tb_second = tb_nch + nn;
    int16_t* ts_node = reinterpret_cast<int16_t*>(dyn + (5 * (size_t)nn + 1) / 2 + 1);   // [NPad] slot -> node (-1 padded)
    int16_t* ts_parent = ts_node + NPad;                                                 // [NPad] slot -> that node's parent
    int16_t* ts_of = ts_parent + NPad;                                                   // [nn] node -> slot (-1: the root)
    double* chain0 = dyn + seg_table_doubles(nn, NPad) + (size_t)cs * seg_chain_doubles(nn, NPad);
    double* Hc = chain0;
    double* Rc = Hc + nn;
    double* Hp = Rc + nn;
    double* Rp = Hp + nn;
    double* tbd = Rp + nn;                                   // summand of node v in the birth-death block, CURRENT state
    double* tcl = tbd + nn;                                  // ... in the clock block
    double* dcur = tcl + nn;                                 // [NPad] distances of the current state
    double* l_dnew = dcur + NPad;                            // [kSegList] the list of this step: new distance, delta, slot
    double* l_delta = l_dnew + kSegList;
    int32_t* l_j = reinterpret_cast<int32_t*>(l_delta + kSegList);
    int32_t* mark = l_j + kSegList;                          // [NPad] the step (+ 1) that last listed the slot
    SegWords* words = reinterpret_cast<SegWords*>(reinterpret_cast<double*>(mark + NPad));
    SegHelpWords* help = reinterpret_cast<SegHelpWords*>(reinterpret_cast<double*>(words) + 8);
    PropApply* A_lds = reinterpret_cast<PropApply*>(reinterpret_cast<double*>(help) + kSegHelpDoubles);
    SegSpec* spec = reinterpret_cast<SegSpec*>(reinterpret_cast<double*>(A_lds) + kSegApplyDoubles);
    lds_vint_t* w_req = lds_vint(&words->req);
    lds_vint_t* w_moves = lds_vint(&words->moves);
    lds_vint_t* w_resp = lds_vint(&words->resp);
    lds_vint_t* w_dec = lds_vint(&words->dec);
    lds_vint_t* w_cnt = lds_vint(&words->cnt);
    lds_vint_t* w_have0 = lds_vint(&words->have0);
    lds_vdouble_t* w_q = lds_vdouble(&words->q);
    lds_vdouble_t* w_lj = lds_vdouble(&words->lj);
    lds_vdouble_t* w_s1 = lds_vdouble(&words->s1);
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    // ---- the tables, by all four waves (the only workgroup barriers of the kernel: before any wave polls a hand-over word)
    const int rr = T.root_right;
    for (int v = threadIdx.x; v < nn; v += NT) {
        tb_parent[v] = M.parent[v];
        tb_size[v] = M.size[v];
        tb_first[v] = P.first_child[v];
        tb_nch[v] = P.n_children[v];
        tb_second[v] = P.second_child[v];
        ts_of[v] = -1;
    }
    if (role == 0 && lane == 0) {
        spec->word = 0;
        help->resp_bd = 0;
        help->resp_cl = 0;
        help->done_bd = 0;
        help->done_cl = 0;
        words->req = 0;
        words->moves = 0;
        words->resp = 0;
        words->dec = 0;
        words->cnt = 0;
        words->have0 = 0;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < NPad; j += NT) {
        const int a = T.slot_node[j];                        // -1 for padded rows
        ts_node[j] = (int16_t)a;
        ts_parent[j] = (int16_t)T.slot_parent[j];
        if (a >= 0) ts_of[a] = (int16_t)j;
    }
    if (threadIdx.x == 0) ts_of[rr] = 0;                     // the root's two daughters share slot 0 (sumFirstTwo); no slot feeds on rr itself
    PriorDev Pst = P;                                        // (the node priors' tables in LDS where they fit)
    if (seg_node_tables_bytes(nn, NPad, P.n_cal, P.n_con) > 0)
        prior_stage_node_tables(Pst, P, dyn + seg_table_doubles(nn, NPad) + 2 * seg_chain_doubles(nn, NPad), (int)threadIdx.x, NT);
    __syncthreads();

    SegChainCtx L;
    L.help = help;
    L.tb_parent = tb_parent;
    L.tb_size = tb_size;
    L.tb_first = tb_first;
    L.tb_nch = tb_nch;
    L.tb_second = tb_second;
    L.Hc = Hc;
    L.Rc = Rc;
    L.Hp = Hp;
    L.Rp = Rp;
    L.tbd = tbd;
    L.tcl = tcl;
    L.words = words;
    L.A_lds = A_lds;
    L.spec = spec;
    L.c = V.c;
    L.logdet = V.logdet;
    // ================================================================ prior waves
    if constexpr (HELP) {
        if (role == 2) {
            seg_prior_wave<0>(M, P, Pst, L, Q, sched, n_steps, step0, seed, b, valid, lane);
            return;
        }
        if (role == 3) {
            seg_prior_wave<1>(M, P, Pst, L, Q, sched, n_steps, step0, seed, b, valid, lane);
            return;
        }
    }
    // ================================================================ likelihood waves
    if (role == 1) {
        double zc[R];                                        // z = L^-1 (d - mu) of the current state, rows 64 k + lane
        double la_pending;
        const bool took = Q.p_acc >= 0 && seg_accept_pending(M, Q, b, seed, la_pending);
        if (took && Q.z_in_zprop) {                          // z' of the accepted dense proposal: copied chain-major (large batches)
#pragma unroll
            for (int k = 0; k < R; ++k) zc[k] = I.zprop[b * I.NPz + 64 * k + lane];
        } else if (took) {                                   // ... or still in the row-split kernel's tiles
            const double* zt = I.zt + ((b >> 4) * I.nr) * 16 + (b & 15);
#pragma unroll
            for (int k = 0; k < R; ++k) zc[k] = (64 * k + lane < I.nr) ? zt[(int64_t)(64 * k + lane) * 16] : 0.0;
        } else {
#pragma unroll
            for (int k = 0; k < R; ++k) zc[k] = I.zcur[b * I.NPz + 64 * k + lane];
        }
        // the current distances (this wave's: it lists the moved ones and commits them) and the slots' marks
        {
            const double* xsrc = (took ? Q.X1 : I.X0) + b * (int64_t)V.n;
            for (int j = lane; j < NPad; j += 64) {
                mark[j] = 0;
                dcur[j] = (j < V.n) ? xsrc[j] : 0.0;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        MhDev Mt = M;                                        // (mh_propose_ranges reads the sub tree sizes and the braces' pointers)
        Mt.size = tb_size;
        // The slots the written nodes of a transform feed -- a written node's own slot, and its daughters' when its height is written -- each
        // once: the first lane to raise the slot's mark to this list's tag lists it.  Slots only: the new distances follow once the proposal
        // is there.  (The transform's integer fields as scalars by value: a struct passed around here ends up in scratch memory.)  Returns the
        // count (the list holds kSegList of them).
        struct SegRanges {
            int kind, hlo, hhi, hlo2, hhi2, rlo, rhi, pt1, pt2, rp1, rp2, rp3, brace_lo, brace_hi;
        };
        auto build_list = [&](int a_kind, int a_hlo, int a_hhi, int a_hlo2, int a_hhi2, int a_rlo, int a_rhi, int a_pt1, int a_pt2, int a_rp1, int a_rp2, int a_rp3, int a_brace_lo, int a_brace_hi, int tagf) __attribute__((always_inline)) -> int {
            const SegRanges A{a_kind, a_hlo, a_hhi, a_hlo2, a_hhi2, a_rlo, a_rhi, a_pt1, a_pt2, a_rp1, a_rp2, a_rp3, a_brace_lo, a_brace_hi};
            int cnt = 0;
            auto emit = [&](bool active, int node_) {        // (every lane calls it: the ballots are the wave's)
                if (__builtin_amdgcn_ballot_w64(active) == 0) return;
                const int slot = active ? (int)ts_of[node_] : -1;
                bool mine = false;
                if (slot >= 0) mine = atomicMax(&mark[slot], tagf) != tagf;      // (the tags count upwards within a launch)
                const uint64_t mk = __builtin_amdgcn_ballot_w64(mine);
                if (mine) {
                    const int pos = cnt + (int)__builtin_popcountll(mk & lt_mask);
                    if (pos < kSegList) l_j[pos] = slot;
                }
                cnt += (int)__builtin_popcountll(mk);
            };
            auto emit_height = [&](bool active, int w) {     // a node whose height is written: its branch and its daughters'
                emit(active, w);
                const int nc = active ? tb_nch[w] : 0;
                emit(nc > 0, active ? tb_first[w] : 0);
                emit(nc > 1, active ? tb_second[w] : 0);
            };
            for (int w0 = A.hlo; w0 < A.hhi; w0 += 64) emit_height(w0 + lane < A.hhi, w0 + lane);
            for (int w0 = A.hlo2; w0 < A.hhi2; w0 += 64) emit_height(w0 + lane < A.hhi2, w0 + lane);
            for (int w0 = A.rlo; w0 < A.rhi; w0 += 64) emit(w0 + lane < A.rhi, w0 + lane);
            {
                // the single nodes in ONE pass: lanes 0 .. 2 the first height-written node with its daughters, 3 .. 5 the second, 6 .. 8
                // the three rate-written ones
                const int g = lane / 3, r = lane - 3 * g;
                int cand = -1;
                if (lane < 6) {
                    const int base = (g == 0) ? A.pt1 : A.pt2;
                    if (base >= 0) cand = (r == 0) ? base : (tb_nch[base] >= r) ? (r == 1 ? tb_first[base] : tb_second[base]) : -1;
                } else if (lane < 9) {
                    cand = (r == 0) ? A.rp1 : (r == 1) ? A.rp2 : A.rp3;
                }
                emit(cand >= 0, cand >= 0 ? cand : 0);
            }
            for (int i = A.brace_lo; i < A.brace_hi; ++i) {
                const int x = M.brace_nodes[i];
                emit_height(lane == 0, x);                   // (SLIDE_BRACE_CONTRA also writes the rates of x and its daughters: the same slots)
            }
            if (cnt > kSegList) cnt = -1;                    // (cannot happen for a proposal mh_capi.cpp put into a segment: the chain wave says so)
            __builtin_amdgcn_s_waitcnt(0xc07f);              // the list is in LDS before any lane reads it
            __builtin_amdgcn_wave_barrier();
            return cnt;
        };
        int p_cur = sched[0];
        const int vz = mh_vzero();                           // (mh_device.hpp: what travels ahead is loaded by vector loads)
        int p_next = sched[(n_steps > 1 ? 1 : 0) + vz];      // (the schedule's entries two steps ahead: a row's loads need its index)
        int kind_cur = M.kind[p_cur], node_cur = M.node[p_cur];
        for (int64_t gs = 0; gs < n_steps; ++gs) {
            const int tag = (int)gs + 1;
            const int p_next2 = sched[((gs + 2 < n_steps) ? gs + 2 : gs) + vz];
            const int kind_next = M.kind[p_next], node_next = M.node[p_next];      // (the next step's row travels while this step computes)
            // ---- AHEAD of the request, while the chain wave draws the proposal: which nodes the proposal writes follows from its table row
            // and the topology alone (mh_propose_ranges), hence the list of moved slots; the first columns of L^-1 they need are touched (one
            // load per column reaches each of its lines: an L2 miss per line now instead of after the request).  The guess is compared with
            // the transform the proposal posts; the rare mismatch (a proposal that bails out on an invalid state) lists again.
            PropApply G;
            mh_propose_ranges(Mt, kind_cur, node_cur, G);
            int c_kind = G.kind, c_hlo = G.hlo, c_hhi = G.hhi, c_hlo2 = G.hlo2, c_hhi2 = G.hhi2, c_rlo = G.rlo, c_rhi = G.rhi, c_pt1 = G.pt1, c_pt2 = G.pt2, c_rp1 = G.rp1, c_rp2 = G.rp2, c_rp3 = G.rp3, c_brace_lo = G.brace_lo, c_brace_hi = G.brace_hi;
            bool c_moves = seg_moves_likelihood(kind_cur, node_cur);
            int cnt = 0;
            constexpr int kPre = (R >= 16) ? 1 : (R >= 12) ? 2 : kSegCols;   // (the register file: one column of 16 doubles per lane at R = 16, four columns below R = 12)
            double pcol[kPre][R];                            // the first columns, requested ahead of the request
            for (int pass = 0; pass < 2; ++pass) {
                cnt = c_moves ? build_list(c_kind, c_hlo, c_hhi, c_hlo2, c_hhi2, c_rlo, c_rhi, c_pt1, c_pt2, c_rp1, c_rp2, c_rp3, c_brace_lo, c_brace_hi, 2 * tag + pass) : 0;
                // the first batch of columns of L^-1 (the list's first slots) is requested now: an L2 miss each, under way while the
                // proposal is still being drawn
                if (cnt > 0) {
#pragma unroll
                    for (int u = 0; u < kPre; ++u) {
                        const int m = (u < cnt) ? u : cnt - 1;
                        const int j = __builtin_amdgcn_readfirstlane(l_j[m]);
                        const double* wc = V.Wc + (size_t)j * NPad + lane;
#pragma unroll
                        for (int k = 0; k < R; ++k) pcol[u][k] = wc[64 * k];
                    }
                }
                if (pass == 1) break;
                (void)seg_poll(w_req, tag, 0);
                const bool moves = *w_moves != 0;
                const int n_kind = A_lds->kind, n_hlo = A_lds->hlo, n_hhi = A_lds->hhi, n_hlo2 = A_lds->hlo2, n_hhi2 = A_lds->hhi2, n_rlo = A_lds->rlo, n_rhi = A_lds->rhi, n_pt1 = A_lds->pt1, n_pt2 = A_lds->pt2, n_rp1 = A_lds->rp1, n_rp2 = A_lds->rp2, n_rp3 = A_lds->rp3, n_brace_lo = A_lds->brace_lo, n_brace_hi = A_lds->brace_hi;
                const bool same_ranges = n_kind == c_kind && n_hlo == c_hlo && n_hhi == c_hhi && n_hlo2 == c_hlo2 && n_hhi2 == c_hhi2 && n_rlo == c_rlo && n_rhi == c_rhi && n_pt1 == c_pt1 && n_pt2 == c_pt2 && n_rp1 == c_rp1 && n_rp2 == c_rp2 && n_rp3 == c_rp3 && n_brace_lo == c_brace_lo && n_brace_hi == c_brace_hi;
                if (moves == c_moves && (!moves || same_ranges)) break;
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
                c_moves = moves;
            }
            // ---- the listed slots' new distances from the proposed state (LDS) and the deltas against the current ones:
            // likelihoodFunctionWrapper, distances = (tH * rMu) * sumFirstTwo (times * rates) (app/Probability.hs:195-207) -- the arithmetic of
            // load_tree (mvn_device.hpp) and of k_mh_step_wg's X1
            if (cnt > 0) {
                const double s1 = *w_s1;
                double d0 = 0.0;                             // the new distance of slot 0, in the lane that holds it
                bool have0 = false;
                for (int m = lane; m < cnt; m += 64) {
                    const int slot = l_j[m];
                    const int a = ts_node[slot], pa = ts_parent[slot];
                    double x = (Hp[pa] - Hp[a]) * Rp[a];
                    if (slot == 0) x = x + (Hp[0] - Hp[rr]) * Rp[rr];
                    x = x * s1;
                    l_dnew[m] = x;
                    l_delta[m] = x - dcur[slot];
                    if (slot == 0) {
                        d0 = x;
                        have0 = true;
                    }
                }
                const uint64_t m0 = __builtin_amdgcn_ballot_w64(have0);
                if (lane == 0) *w_have0 = (m0 != 0) ? 1 : 0;
                if (m0 != 0) {
                    const double lj1 = log(1.0 / mh_readlane64(d0, (int)__builtin_ctzll(m0)));     // jacobianRootBranch, :393-410
                    if (lane == 0) *w_lj = lj1;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);          // the list's values are in LDS before any lane reads them
                __builtin_amdgcn_wave_barrier();
            } else if (lane == 0) {
                *w_have0 = 0;
            }
            double zp[R];
#pragma unroll
            for (int k = 0; k < R; ++k) zp[k] = zc[k];
            // kSegCols columns in flight: a column is an L2 miss (W is 8 MB at N = 1023), a batch costs its latency once
            if (cnt > 0) {                                   // the first batch: the columns requested ahead of the request
#pragma unroll
                for (int u = 0; u < kPre; ++u) {
                    const double dl = (u < cnt) ? l_delta[u] : 0.0;    // (past the end: the last column again with weight 0: exact)
#pragma unroll
                    for (int k = 0; k < R; ++k) zp[k] = fma(dl, pcol[u][k], zp[k]);
                }
            }
            for (int m0 = kPre; m0 < cnt; m0 += kSegCols) {
                double col[kSegCols][R], dl[kSegCols];
#pragma unroll
                for (int u = 0; u < kSegCols; ++u) {
                    const int m = (m0 + u < cnt) ? m0 + u : cnt - 1;   // (past the end: the last column again with weight 0: exact)
                    const int j = __builtin_amdgcn_readfirstlane(l_j[m]);
                    dl[u] = (m0 + u < cnt) ? l_delta[m] : 0.0;
                    const double* wc = V.Wc + (size_t)j * NPad + lane;
#pragma unroll
                    for (int k = 0; k < R; ++k) col[u][k] = wc[64 * k];
                }
#pragma unroll
                for (int u = 0; u < kSegCols; ++u)
#pragma unroll
                    for (int k = 0; k < R; ++k) zp[k] = fma(dl[u], col[u][k], zp[k]);
            }
            double sq = 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) sq = fma(zp[k], zp[k], sq);
            const double q = wave_sum(sq);
            if (lane == 0) {
                *w_q = q;
                *w_cnt = cnt;
            }
            seg_post(w_resp, tag);                           // (every lane stores the same word: the fence is the wave's)
            const int d = seg_poll(w_dec, tag, 1);
            if ((d & 1) && cnt > 0) {
#pragma unroll
                for (int k = 0; k < R; ++k) zc[k] = zp[k];
                for (int m = lane; m < cnt; m += 64) dcur[l_j[m]] = l_dnew[m];
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
            p_cur = p_next;
            p_next = p_next2;
            kind_cur = __builtin_amdgcn_readfirstlane(kind_next);
            node_cur = __builtin_amdgcn_readfirstlane(node_next);
        }
        if (valid) {
#pragma unroll
            for (int k = 0; k < R; ++k) I.zcur[b * I.NPz + 64 * k + lane] = zc[k];
            for (int j = lane; j < V.n; j += 64) I.X0[b * (int64_t)V.n + j] = dcur[j];
        }
        seg_tail_distances(M, Q, words, Hp, Rp, ts_node, ts_parent, rr, V.n, n_steps, b, valid, lane);
        return;
    }

    // ================================================================ chain waves (mh_segment_device.hpp: shared with the sparse kernel)
    seg_chain_wave<HELP, false>(M, P, Pst, L, Q, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, b, valid, lane);
}

// trees whose factor takes 6 .. 16 register blocks (259 .. 1026 nodes: below that the streaming chain kernel's in-kernel sweeps of
// the dense proposals cost less than two launches), columns of L^-1 on the device, tables and two chains within a CU's LDS
bool mh_segment_available(const MhDev& M, const MvnDev& V)
{
    if ((V.R != 6 && V.R != 8 && V.R != 12 && V.R != 16) || V.Wc == nullptr || M.n_nodes > 64 * V.R + 2 || M.n_nodes < 3 || M.batch > kSegMaxBatch) return false;
    return seg_lds_bytes(M.n_nodes, 64 * V.R) <= 160 * 1024;
}

template <int R, bool HELP>
static hipError_t launch_segment_RH(const MhDev& M, const MvnDev& V, const TreeDev& T, const PriorDev& P, const MhInc& I, const int32_t* sched,
                                   int64_t n_steps, int32_t S, int accumulate, uint64_t step0, uint64_t seed, double* trace_alpha,
                                   int8_t* trace_accept, int64_t gs_base, int summands_kept, const MhSegPending& Q, hipStream_t st)
{
    const size_t dynb = seg_lds_bytes(M.n_nodes, 64 * R) + seg_node_tables_bytes(M.n_nodes, 64 * R, P.n_cal, P.n_con);
    static std::atomic<unsigned long long> allowed{0};       // more than 64 KiB of LDS has to be allowed once per device
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!((allowed.load(std::memory_order_acquire) >> dev) & 1ull)) {
        if (hipError_t e = hipFuncSetAttribute((const void*)k_mh_segment<R, HELP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) return e;
        allowed.fetch_or(1ull << dev, std::memory_order_release);
    }
    note_dynamic_lds(dynb);
    hipLaunchKernelGGL((k_mh_segment<R, HELP>), dim3((unsigned)((M.batch + 1) / 2)), dim3(HELP ? 512 : 256), dynb, st, M, V, T, P, I, sched, n_steps, S, accumulate, step0,
                       seed, trace_alpha, trace_accept, gs_base, summands_kept, Q);
    return hipGetLastError();
}

template <int R>
static hipError_t launch_segment_R(const MhDev& M, const MvnDev& V, const TreeDev& T, const PriorDev& P, const MhInc& I, const int32_t* sched,
                                   int64_t n_steps, int32_t S, int accumulate, uint64_t step0, uint64_t seed, double* trace_alpha,
                                   int8_t* trace_accept, int64_t gs_base, int summands_kept, const MhSegPending& Q, hipStream_t st)
{
    // (mcd_set_option "MCD_MH_PRIOR_WAVES" = 0: the chain wave evaluates the whole ln prior; tests, timing)
    if (opt_is(OPT_MH_PRIOR_WAVES, 0))
        return launch_segment_RH<R, false>(M, V, T, P, I, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, st);
    return launch_segment_RH<R, true>(M, V, T, P, I, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, st);
}

// steps [0, n_steps) of `sched` (device memory), none of which moves more than kSegList distances; step0 = the step number of
// sched[0], gs_base its position in the run's schedule (iterations close at multiples of S); summands_kept: MhDev::psum holds the
// current states' summands; pending (may be null): a dense proposal whose likelihood has been evaluated and which is decided here
hipError_t launch_mh_segment(const MhDev& M, const MvnDev& V, const TreeDev& T, const PriorDev& P, const MhInc& I, const int32_t* sched,
                             int64_t n_steps, int32_t S, int accumulate, uint64_t step0, uint64_t seed, double* trace_alpha, int8_t* trace_accept,
                             int64_t gs_base, int summands_kept, const MhSegPending* pending, hipStream_t st)
{
    MhSegPending Q{};
    Q.p_acc = -1;
    Q.p_tail = -1;
    if (pending) Q = *pending;
    Q.ahead_from = opt_or(OPT_MH_AHEAD_FROM, kSegAheadFrom);
    Q.prior_draws = (!opt_is(OPT_MH_PRIOR_DRAWS, 0) && !opt_is(OPT_MH_PRIOR_WAVES, 0)) ? 1 : 0;
    if (n_steps <= 0) return Q.p_acc >= 0 ? hipErrorInvalidValue : hipSuccess;
    if (Q.p_acc >= 0 && (Q.X1 == nullptr || (Q.z_in_zprop ? I.zprop == nullptr : I.zt == nullptr) || !summands_kept)) return hipErrorInvalidValue;
    if (Q.p_tail >= M.n_prop || (Q.p_tail >= 0 && (Q.X1_tail == nullptr || M.psum == nullptr || M.psel == nullptr))) return hipErrorInvalidValue;
    if (n_steps > (1 << 28)) return hipErrorInvalidValue;    // (the hand-over words count steps in 30 bits)
    if (!mh_segment_available(M, V) || I.X0 == nullptr || I.zcur == nullptr || I.NPz != 64 * V.R) return hipErrorInvalidValue;
    if (V.R == 6) return launch_segment_R<6>(M, V, T, P, I, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, st);
    if (V.R == 8) return launch_segment_R<8>(M, V, T, P, I, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, st);
    if (V.R == 12) return launch_segment_R<12>(M, V, T, P, I, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, st);
    return launch_segment_R<16>(M, V, T, P, I, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, st);
}

}  // namespace mcd
