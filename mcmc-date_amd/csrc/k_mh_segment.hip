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
#include "mh_device.hpp"
#include "prior_device.hpp"

#include <atomic>

namespace mcd {

constexpr int kSegApplyDoubles = (int)((sizeof(PropApply) + 7) / 8);   // the proposal's per-node transform, chain wave -> likelihood wave
#ifndef MCD_SEG_COLS
#define MCD_SEG_COLS 4
#endif
constexpr int kSegCols = MCD_SEG_COLS;   // columns of L^-1 in flight per batch
constexpr int64_t kSegMaxBatch = 65536;   // (beyond 512 chains the workgroups run in rounds: a launch lasts rounds x steps)
constexpr int kSegList = kMhSegList;   // moved distances of one proposal at most (mh_capi.cpp: proposals that may move more are dense)

__device__ __forceinline__ bool seg_moves_likelihood(int kind, int node)
{
    return !(kind == MCD_PROP_SCALE_SCALAR && (node == 0 || node == 1 || node == 4));
}

// LDS, in doubles.  Shared by the two chains: five int32 tables of the tree (parent, sub tree size, first / second child, number
// of children), and three int16 ones of the distance slots (slot -> node, slot -> that node's parent, node -> slot).
__host__ __device__ inline size_t seg_table_doubles(int n_nodes, int np) { return (5 * (size_t)n_nodes + 1) / 2 + 1 + ((size_t)n_nodes + 2 * (size_t)np + 3) / 4 + 1; }
// Per chain: 4 state rows, the summands of the two blocks, the current distances [np]; the list (new distance, delta: doubles; slot:
// int32); the slots' marks (int32 [np]); eight words of hand-over; the proposal's per-node transform.
__host__ __device__ inline size_t seg_chain_doubles(int n_nodes, int np)
{
    return 6 * (size_t)n_nodes + (size_t)np + 2 * (size_t)kSegList + (size_t)kSegList / 2 + (size_t)np / 2 + 8 + (size_t)kSegApplyDoubles;
}

__host__ __device__ inline size_t seg_lds_bytes(int n_nodes, int np) { return sizeof(double) * (seg_table_doubles(n_nodes, np) + 2 * seg_chain_doubles(n_nodes, np)); }
// ... and, where it still fits, the calibration and constraint tables behind them (prior_device.hpp: prior_stage_node_tables)
__host__ __device__ inline size_t seg_node_tables_bytes(int n_nodes, int np, int n_cal, int n_con)
{
    const size_t need = sizeof(double) * prior_node_tables_doubles(n_cal, n_con);
    return (need > 0 && seg_lds_bytes(n_nodes, np) + need <= 160 * 1024) ? need : 0;
}

struct SegWords {                  // the hand-over between a chain's two waves (LDS)
    int req;                       // chain wave: step + 1 when the proposal of step `step` is applied (Hp, Rp) and its transform posted
    int moves;                     // ... whether it can move the distances at all
    int resp;                      // likelihood wave: step + 1 when q is there
    int dec;                       // chain wave: 2 (step + 1) + accepted
    int cnt;                       // likelihood wave: moved distances (-1: more than the list holds)
    int have0;                     // ... whether slot 0 is among them (then lj = ln jacobianRootBranch of the proposal)
    double q;                      // |z'|^2
    double lj;
    double s1;                     // chain wave: tH * rMu of the proposal
    double pad[2];
};
static_assert(sizeof(SegWords) == 64, "eight doubles of LDS");

__device__ __forceinline__ int seg_poll(lds_vint_t* w, int want_shifted, int shift)
{
    int v = *w;
    while ((v >> shift) != want_shifted) {
        __builtin_amdgcn_s_sleep(1);
        v = *w;
    }
    lds_acquire_fence();                                     // what the other wave wrote before the word is read after it
    return v;
}
// publish: everything this wave wrote to LDS so far is there before the word is
__device__ __forceinline__ void seg_post(lds_vint_t* w, int value)
{
    lds_publish_fence();
    *w = value;
}

// Does the pending DENSE proposal of chain b (proposed by k_mh_step_wg, its ln likelihood by the row-split launch) become the
// current state?  The decision of k_mh_step_wg's accept half, bit for bit; both waves of a chain take it for themselves.
__device__ __forceinline__ bool seg_accept_pending(const MhDev& M, const MhSegPending& Q, int64_t b, uint64_t seed, double& la_out)
{
    const int64_t B = M.batch;
    const double lp = M.post[b], ll = M.post[B + b], lj = M.post[2 * B + b];
    const double lp1 = M.post1[b], ll1 = M.post1[B + b], lj1 = M.post1[2 * B + b];
    double la = M.beta[b] * ((lp1 + ll1) - (lp + ll)) + M.lnqj[b];
    if (Q.jac_root) la += (double)Q.jac_root * (lj1 - lj);
    double ua, ub;
    philox_block(mh_rng(seed, M.chain0 + b, Q.step), 0xFFFFFFFFu, ua, ub);
    la_out = la;
    return (la >= 0) || (ua < exp(la));
}

template <int R>
__global__ __launch_bounds__(256, 1) void k_mh_segment(MhDev M, MvnDev V, TreeDev T, PriorDev P, MhInc I, const int32_t* __restrict__ sched,
                                                    int64_t n_steps, int32_t S, int accumulate, uint64_t step0, uint64_t seed,
                                                    double* __restrict__ trace_alpha, int8_t* __restrict__ trace_accept, int64_t gs_base,
                                                    int summands_kept, MhSegPending Q)
{
    extern __shared__ double dyn[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cs = wave & 1;                                 // which of the workgroup's two chains
    const int nn = M.n_nodes, NPr = M.n_prop;
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
    PropApply* A_lds = reinterpret_cast<PropApply*>(reinterpret_cast<double*>(words) + 8);
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
    for (int v = threadIdx.x; v < nn; v += 256) {
        tb_parent[v] = M.parent[v];
        tb_size[v] = M.size[v];
        tb_first[v] = P.first_child[v];
        tb_nch[v] = P.n_children[v];
        tb_second[v] = P.second_child[v];
        ts_of[v] = -1;
    }
    if (wave < 2 && lane == 0) {
        words->req = 0;
        words->moves = 0;
        words->resp = 0;
        words->dec = 0;
        words->cnt = 0;
        words->have0 = 0;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < NPad; j += 256) {
        const int a = T.slot_node[j];                        // -1 for padded rows
        ts_node[j] = (int16_t)a;
        ts_parent[j] = (int16_t)T.slot_parent[j];
        if (a >= 0) ts_of[a] = (int16_t)j;
    }
    if (threadIdx.x == 0) ts_of[rr] = 0;                     // the root's two daughters share slot 0 (sumFirstTwo); no slot feeds on rr itself
    PriorDev Pst = P;                                        // (the node priors' tables in LDS where they fit)
    if (seg_node_tables_bytes(nn, NPad, P.n_cal, P.n_con) > 0)
        prior_stage_node_tables(Pst, P, dyn + seg_table_doubles(nn, NPad) + 2 * seg_chain_doubles(nn, NPad), (int)threadIdx.x, 256);
    __syncthreads();

    // ================================================================ likelihood waves
    if (wave >= 2) {
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
        for (int64_t gs = 0; gs < n_steps; ++gs) {
            const int tag = (int)gs + 1;
            (void)seg_poll(w_req, tag, 0);
            // ---- the distances the written nodes feed: a written node's own slot, and its daughters' when its height is written.  Each
            // slot once (the first lane to exchange the slot's mark for this step's lists it), with its new distance and the delta.
            int cnt = 0;
            if (*w_moves) {
                const PropApply A = *A_lds;
                const double s1 = *w_s1;
                double d0 = 0.0;                             // the new distance of slot 0, in the lane that listed it
                bool have0 = false;
                auto emit = [&](bool active, int node_) {    // (every lane calls it: the ballots are the wave's)
                    if (__builtin_amdgcn_ballot_w64(active) == 0) return;
                    const int slot = active ? (int)ts_of[node_] : -1;
                    bool mine = false;
                    if (slot >= 0) mine = atomicExch(&mark[slot], tag) != tag;
                    const uint64_t mk = __builtin_amdgcn_ballot_w64(mine);
                    if (mine) {
                        const int pos = cnt + (int)__builtin_popcountll(mk & lt_mask);
                        // likelihoodFunctionWrapper: distances = (tH * rMu) * sumFirstTwo (times * rates)   (app/Probability.hs:195-207), the
                        // arithmetic of load_tree (mvn_device.hpp) and of k_mh_step_wg's X1
                        const int a = ts_node[slot], pa = ts_parent[slot];
                        double x = (Hp[pa] - Hp[a]) * Rp[a];
                        if (slot == 0) x = x + (Hp[0] - Hp[rr]) * Rp[rr];
                        x = x * s1;
                        if (pos < kSegList) {
                            l_j[pos] = slot;
                            l_dnew[pos] = x;
                            l_delta[pos] = x - dcur[slot];
                        }
                        if (slot == 0) {
                            d0 = x;
                            have0 = true;
                        }
                    }
                    cnt += (int)__builtin_popcountll(mk);
                };
                auto emit_height = [&](bool active, int w) {    // a node whose height is written: its branch and its daughters'
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
                    emit_height(lane == 0, x);               // (SLIDE_BRACE_CONTRA also writes the rates of x and its daughters: the same slots)
                }
                const uint64_t m0 = __builtin_amdgcn_ballot_w64(have0);
                if (lane == 0) *w_have0 = (m0 != 0) ? 1 : 0;
                if (m0 != 0) {
                    const double lj1 = log(1.0 / mh_readlane64(d0, (int)__builtin_ctzll(m0)));     // jacobianRootBranch, :393-410
                    if (lane == 0) *w_lj = lj1;
                }
                if (cnt > kSegList) cnt = -1;                // (cannot happen for a proposal mh_capi.cpp put into a segment: the chain wave says so)
                __builtin_amdgcn_s_waitcnt(0xc07f);          // the list is in LDS before any lane reads it
                __builtin_amdgcn_wave_barrier();
            }
            double zp[R];
#pragma unroll
            for (int k = 0; k < R; ++k) zp[k] = zc[k];
            // kSegCols columns in flight: a column is an L2 miss (W is 8 MB at N = 1023), a batch costs its latency once
            for (int m0 = 0; m0 < cnt; m0 += kSegCols) {
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
        }
        if (valid) {
#pragma unroll
            for (int k = 0; k < R; ++k) I.zcur[b * I.NPz + 64 * k + lane] = zc[k];
            for (int j = lane; j < V.n; j += 64) I.X0[b * (int64_t)V.n + j] = dcur[j];
        }
        return;
    }

    // ================================================================ chain waves
    // a pending dense proposal is decided here (instead of by a launch of k_mh_step_wg that would do nothing else): the chain then
    // starts from the proposed state
    double la_pending = 0.0;
    const bool took = Q.p_acc >= 0 && seg_accept_pending(M, Q, b, seed, la_pending);
    if (Q.p_acc >= 0 && lane == 0 && valid) {
        atomicAdd(&M.tried[b * NPr + Q.p_acc], 1);
        if (took) atomicAdd(&M.acc[b * NPr + Q.p_acc], 1);
        if (Q.trace_alpha) Q.trace_alpha[b] = la_pending;
        if (Q.trace_accept) Q.trace_accept[b] = took ? 1 : 0;
    }
    MhDev Ml = M;
    Ml.parent = tb_parent;
    Ml.size = tb_size;
    PriorDev Pl = Pst;
    Pl.parent = tb_parent;
    Pl.first_child = tb_first;
    Pl.n_children = tb_nch;
    Pl.second_child = tb_second;
    const double* tune = M.tune + b * NPr;                   // (constant during a launch: mcd_mh_tune is a call of its own)
    int32_t* acc = M.acc + b * NPr;
    int32_t* tried = M.tried + b * NPr;
    const double* Hsrc = (took ? M.H1 : M.H) + b * M.ld;
    const double* Rsrc = (took ? M.R1 : M.R) + b * M.ld;
    for (int w = lane; w < nn; w += 64) {
        const double h = Hsrc[w], r = Rsrc[w];
        Hc[w] = h;
        Rc[w] = r;
        Hp[w] = h;                                           // invariant between steps: proposed arrays = current arrays
        Rp[w] = r;
    }
    double sc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) sc[i] = (took ? M.sc1 : M.sc)[i * B + b];
    const double* psrc = took ? M.post1 : M.post;
    double lp = psrc[b], ll = psrc[B + b], lj = psrc[2 * B + b];
    const double beta = M.beta[b];
    __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): this wave's LDS writes have landed (one wave: in order)
    __builtin_amdgcn_wave_barrier();
    // the three blocks of the ln prior of the current state and the summands of two of them: from the step kernel's kept ones
    // (summand of node v at v - 1), or evaluated here
    ClockCache cc{__builtin_nan(""), 0.0, 0.0, 0.0};
    double c0, c1, c2;
    const bool from_kept = summands_kept && M.psum != nullptr;
    int seg_sel = 0;
    if (from_kept) {
        const int2 sp = reinterpret_cast<const int2*>(M.psel)[b];
        seg_sel = took ? (sp.x ^ ((sp.y >> 1) & 3)) : sp.x;  // (the accepted proposal's blocks are the current ones)
    }
    const size_t NS = (size_t)((nn - 1 + 63) / 64) * 64;
    if (from_kept) {
        const double* s_bd = M.psum + ((size_t)b * 4 + (size_t)(seg_sel & 1)) * NS;
        const double* s_cl = M.psum + ((size_t)b * 4 + 2 + (size_t)((seg_sel >> 1) & 1)) * NS;
        for (int v = 1 + lane; v < nn; v += 64) {
            tbd[v] = s_bd[v - 1];
            tcl[v] = s_cl[v - 1];
        }
        const double* pc = (took ? M.pcomp1 : M.pcomp) + b * 3;
        c0 = pc[0];
        c1 = pc[1];
        c2 = pc[2];
        prior_clock_scalars(sc[4], cc);
    } else {
        c0 = prior_nodes_wave(Pl, lane, sc[2], Hc);
        const bool near = prior_bd_near(sc[0], sc[1]);
        double bd = 0.0, cl = 0.0;
        prior_clock_scalars(sc[4], cc);
        for (int v = 1 + lane; v < nn; v += 64) {
            const double t1 = prior_bd_term(Pl, v, near, sc[0], sc[1], Hc);
            const double t2 = prior_clock_term(Pl, v, sc[4], cc.lg_k, cc.log_t, Hc, Rc);
            tbd[v] = t1;
            tcl[v] = t2;
            bd += t1;
            cl += t2;
        }
        c1 = prior_bd_finish(pr_wave_sum(bd), sc[0], sc[1]);
        c2 = prior_clock_finish(Pl, pr_wave_sum(cl), sc[3], sc[4], cc.hyper);
        lp = c0 + c1 + c2;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (Q.p_acc >= 0 && Q.accumulate && valid) {             // the pending step closed an iteration of the cycle
        for (int w = lane; w < nn; w += 64) {
            const double a = sc[2] * Hc[w];
            M.age_sum[b * nn + w] += a;
            M.age_sq[b * nn + w] += a * a;
        }
    }
    // The nodes a proposal writes (PropApply: up to three pre-order ranges, five single nodes, the braced nodes with their
    // daughters): f(w) for each of them, lanes in parallel (a node may come twice).  Between steps the proposed arrays equal the
    // current ones, so a step applies, commits or takes back its proposal on these nodes only.
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
#ifdef MCD_SEG_STAMP
    // diagnostic build (make stamp_seg): s_memtime ticks per phase of the chain wave, summed over the launch, in the first rows of
    // trace_alpha: 0 loop head + draws, 1 propose, 2 list of moved distances, 3 ln prior, 4 waiting for |z'|^2, 5 decision + commit
    uint64_t tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SEG_TICK(i)                                       \
    {                                                     \
        const uint64_t now_ = __builtin_readcyclecounter(); \
        tk[i] += now_ - t_last;                           \
        t_last = now_;                                    \
    }
    uint64_t t_last = __builtin_readcyclecounter();
#else
#define SEG_TICK(i)
#endif
    int p = sched[0];
    PropRow row = mh_load_row(M, p);
    double t_cur = tune[p];
    StepDraws pre{1.0, 0.0, 0.0, 0.5, 0.5};                  // lane l: the state-independent draws of step (gs & ~63) + l
    for (int64_t gs = 0; gs < n_steps; ++gs) {
        const int tag = (int)gs + 1;
        const int p_next = (gs + 1 < n_steps) ? sched[gs + 1] : p;
        const PropRow row_next = mh_load_row(M, p_next);     // the next step's row travels while this step computes
        const double t_next = tune[p_next];                  // ... and its tuning parameter (global memory: a load at the point of use stalled the proposal)
        if ((gs & 63) == 0) {
            // 64 consecutive steps at once, one step per lane: what can be drawn knowing only the proposal row and its tuning
            // parameter (as k_mh_draws does for the two-launch path)
            const int64_t mine = gs + lane;
            if (mine < n_steps) {
                const int pl = sched[mine];
                pre = mh_step_draws(mh_load_row(M, pl), tune[pl], mh_rng(seed, M.chain0 + b, step0 + (uint64_t)mine));
            }
        }
        const int sl = (int)(gs & 63);
        const StepDraws dr{mh_readlane64(pre.u, sl), mh_readlane64(pre.lnq, sl), mh_readlane64(pre.logu, sl), mh_readlane64(pre.U, sl),
                           mh_readlane64(pre.Uacc, sl)};
        double sc1[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) sc1[i] = sc[i];
        SEG_TICK(0)
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
        SEG_TICK(1)
        // ---- the likelihood wave takes it from here: the transform (which nodes are written), Hp / Rp, tH * rMu
        const bool moves = seg_moves_likelihood(row.kind, row.node);
        double lj1 = lj;
        if (lane == 0) {
            *A_lds = A;
            *w_s1 = sc1[2] * sc1[3];
            *w_moves = moves ? 1 : 0;
        }
        seg_post(w_req, tag);                                // (every lane stores the same word: the fence is the wave's)
        SEG_TICK(2)
        // ---- ln prior: only the blocks whose inputs the proposal writes (a superset of "changed": a block re-evaluated on unchanged
        // inputs returns the same bits)
        const bool dH = A.hhi > A.hlo || A.hhi2 > A.hlo2 || A.pt1 >= 0 || A.pt2 >= 0 || A.brace_hi > A.brace_lo;
        const bool dR = A.rhi > A.rlo || A.rp1 >= 0 || A.rp2 >= 0 || A.rp3 >= 0 || (A.brace_hi > A.brace_lo && A.kind == MCD_PROP_SLIDE_BRACE_CONTRA);
        ClockCache ccp = cc;                                 // refreshed only if the proposal moved rVar
        const double c0p = (dH || sc1[2] != sc[2]) ? prior_nodes_wave(Pl, lane, sc1[2], Hp) : c0;
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
        double old_bd = 0.0, old_cl = 0.0;                   // the summands this lane overwrites, until the decision
        const int v_bd = few_bd ? cand_bd(lane) : -1;
        const bool mine_bd = few_bd && lane < cnt_bd && v_bd >= 1;
        if (few_bd) {
            // in place (a node may come twice: every lane reads the old value before any lane writes -- LDS keeps a wave's order)
            if (mine_bd) old_bd = tbd[v_bd];
            const double t = mine_bd ? prior_bd_term(Pl, v_bd, false, sc1[0], sc1[1], Hp) : 0.0;
            if (mine_bd) tbd[v_bd] = t;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            double bd = 0.0;
            for (int w = 1 + lane; w < nn; w += 64) bd += tbd[w];
            c1p = prior_bd_finish(pr_wave_sum(bd), sc1[0], sc1[1]);
        } else if (need_bd) {
            // every summand (the rates moved, or many heights): the sum alone; the summands are evaluated again if the proposal is accepted
            const bool near = prior_bd_near(sc1[0], sc1[1]);
            double bd = 0.0;
            for (int v = 1 + lane; v < nn; v += 64) bd += prior_bd_term(Pl, v, near, sc1[0], sc1[1], Hp);
            c1p = prior_bd_finish(pr_wave_sum(bd), sc1[0], sc1[1]);
        }
        const bool cl_heights = dH && P.clock_model >= 2;    // white noise / autocorrelated: the summands also hold branch durations
        const bool need_cl = dR || sc1[3] != sc[3] || sc1[4] != sc[4] || cl_heights;
        const bool few_cl = need_cl && sc1[4] == sc[4] && P.clock_model < 2 && cnt_cl <= 64;
        double c2p = c2;
        const int v_cl = few_cl ? cand_cl(lane) : -1;
        const bool mine_cl = few_cl && lane < cnt_cl && v_cl >= 1;
        if (few_cl) {
            if (mine_cl) old_cl = tcl[v_cl];
            const double t = mine_cl ? prior_clock_term(Pl, v_cl, sc1[4], cc.lg_k, cc.log_t, Hp, Rp) : 0.0;
            if (mine_cl) tcl[v_cl] = t;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            double cl = 0.0;
            for (int w = 1 + lane; w < nn; w += 64) cl += tcl[w];
            c2p = prior_clock_finish(Pl, pr_wave_sum(cl), sc1[3], sc1[4], cc.hyper);
        } else if (need_cl) {
            if (ccp.va != sc1[4]) prior_clock_scalars(sc1[4], ccp);
            double cl = 0.0;
            for (int v = 1 + lane; v < nn; v += 64) cl += prior_clock_term(Pl, v, sc1[4], ccp.lg_k, ccp.log_t, Hp, Rp);
            c2p = prior_clock_finish(Pl, pr_wave_sum(cl), sc1[3], sc1[4], ccp.hyper);
        }
        const double lp1 = c0p + c1p + c2p;
        SEG_TICK(3)
        // ---- the likelihood wave's answer
        (void)seg_poll(w_resp, tag, 0);
        SEG_TICK(4)
        double ll1 = ll;
        if (moves) {
            const double q = *w_q;
            const int cnt = *w_cnt;
            if (*w_have0) lj1 = *w_lj;
            if (cnt < 0) ll = __builtin_nan("");             // more moved distances than the list holds -- cannot happen for a proposal mh_capi.cpp
                                                             // put into a segment; if it does, the chain says so: its ln likelihood is NaN from here
                                                             // on (nothing is accepted, mcd_mh_get_posterior shows it), not a silently wrong value
            ll1 = (cnt < 0) ? __builtin_nan("") : V.c + (-0.5) * (V.logdet + q);      // :169 (finish_ll)
        }
        double la = beta * ((lp1 + ll1) - (lp + ll)) + lnqj;           // heated chains of MC3: posterior^beta; beta = 1 is exact
        if (row.jac_root) la += (double)row.jac_root * (lj1 - lj);
        const bool ok = (la >= 0) || (dr.Uacc < exp(la));
        seg_post(w_dec, 2 * tag + (ok ? 1 : 0));
        if (ok) {
            for_write_set(A, [&](int w) {
                Hc[w] = Hp[w];
                Rc[w] = Rp[w];
            });
#pragma unroll
            for (int i = 0; i < 5; ++i) sc[i] = sc1[i];
            // (few: already in place)  every summand: evaluated again, now to be kept -- the same function results as the sum's
            if (!few_bd && need_bd) {
                const bool near = prior_bd_near(sc1[0], sc1[1]);
                for (int v = 1 + lane; v < nn; v += 64) tbd[v] = prior_bd_term(Pl, v, near, sc1[0], sc1[1], Hp);
            }
            if (!few_cl && need_cl)
                for (int v = 1 + lane; v < nn; v += 64) tcl[v] = prior_clock_term(Pl, v, sc1[4], ccp.lg_k, ccp.log_t, Hp, Rp);
            c0 = c0p;
            c1 = c1p;
            c2 = c2p;
            cc = ccp;
            lp = lp1;
            ll = ll1;
            lj = lj1;
        } else {
            for_write_set(A, [&](int w) {
                Hp[w] = Hc[w];
                Rp[w] = Rc[w];
            });
            if (mine_bd) tbd[v_bd] = old_bd;                 // the overwritten summands back
            if (mine_cl) tcl[v_cl] = old_cl;
        }
        if (lane == 0 && valid) {
            atomicAdd(&tried[p], 1);                         // (global memory, no value returned: nothing waits for it)
            if (ok) atomicAdd(&acc[p], 1);
            if (trace_alpha) trace_alpha[gs * B + b] = la;
            if (trace_accept) trace_accept[gs * B + b] = ok ? 1 : 0;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (accumulate && valid && (gs_base + gs + 1) % S == 0) {      // once per iteration of the cycle: straight into the running sums
            for (int w = lane; w < nn; w += 64) {
                const double a = sc[2] * Hc[w];
                M.age_sum[b * nn + w] += a;
                M.age_sq[b * nn + w] += a * a;
            }
        }
        p = p_next;
        row = row_next;
        t_cur = t_next;
        SEG_TICK(5)
    }
#ifdef MCD_SEG_STAMP
    if (trace_alpha && lane == 0 && valid)
        for (int i = 0; i < 8; ++i) trace_alpha[(int64_t)i * B + b] = (double)tk[i];
#endif
    if (!valid) return;
    // ---- back to where the two-launch path keeps a chain
    for (int w = lane; w < nn; w += 64) {
        M.H[b * M.ld + w] = Hc[w];
        M.R[b * M.ld + w] = Rc[w];
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) M.sc[i * B + b] = sc[i];
        M.post[b] = lp;
        M.post[B + b] = ll;
        M.post[2 * B + b] = lj;
        M.pcomp[b * 3 + 0] = c0;
        M.pcomp[b * 3 + 1] = c1;
        M.pcomp[b * 3 + 2] = c2;
    }
    if (M.psum != nullptr) {
        double* s_bd = M.psum + ((size_t)b * 4 + (size_t)(seg_sel & 1)) * NS;
        double* s_cl = M.psum + ((size_t)b * 4 + 2 + (size_t)((seg_sel >> 1) & 1)) * NS;
        for (int v = 1 + lane; v < nn; v += 64) {
            s_bd[v - 1] = tbd[v];
            s_cl[v - 1] = tcl[v];
        }
        if (lane == 0) reinterpret_cast<int2*>(M.psel)[b] = make_int2(seg_sel, 0);
    }
}

// trees whose factor takes 6 .. 16 register blocks (259 .. 1026 nodes: below that the streaming chain kernel's in-kernel sweeps of
// the dense proposals cost less than two launches), columns of L^-1 on the device, tables and two chains within a CU's LDS
bool mh_segment_available(const MhDev& M, const MvnDev& V)
{
    if ((V.R != 6 && V.R != 8 && V.R != 12 && V.R != 16) || V.Wc == nullptr || M.n_nodes > 64 * V.R + 2 || M.n_nodes < 3 || M.batch > kSegMaxBatch) return false;
    return seg_lds_bytes(M.n_nodes, 64 * V.R) <= 160 * 1024;
}

template <int R>
static hipError_t launch_segment_R(const MhDev& M, const MvnDev& V, const TreeDev& T, const PriorDev& P, const MhInc& I, const int32_t* sched,
                                   int64_t n_steps, int32_t S, int accumulate, uint64_t step0, uint64_t seed, double* trace_alpha,
                                   int8_t* trace_accept, int64_t gs_base, int summands_kept, const MhSegPending& Q, hipStream_t st)
{
    const size_t dynb = seg_lds_bytes(M.n_nodes, 64 * R) + seg_node_tables_bytes(M.n_nodes, 64 * R, P.n_cal, P.n_con);
    static std::atomic<unsigned long long> allowed{0};       // more than 64 KiB of LDS has to be allowed once per device
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!((allowed.load(std::memory_order_acquire) >> dev) & 1ull)) {
        if (hipError_t e = hipFuncSetAttribute((const void*)k_mh_segment<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) return e;
        allowed.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL(k_mh_segment<R>, dim3((unsigned)((M.batch + 1) / 2)), dim3(256), dynb, st, M, V, T, P, I, sched, n_steps, S, accumulate, step0,
                       seed, trace_alpha, trace_accept, gs_base, summands_kept, Q);
    return hipGetLastError();
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
    if (pending) Q = *pending;
    if (n_steps <= 0) return Q.p_acc >= 0 ? hipErrorInvalidValue : hipSuccess;
    if (Q.p_acc >= 0 && (Q.X1 == nullptr || (Q.z_in_zprop ? I.zprop == nullptr : I.zt == nullptr) || !summands_kept)) return hipErrorInvalidValue;
    if (n_steps > (1 << 28)) return hipErrorInvalidValue;    // (the hand-over words count steps in 30 bits)
    if (!mh_segment_available(M, V) || I.X0 == nullptr || I.zcur == nullptr || I.NPz != 64 * V.R) return hipErrorInvalidValue;
    if (V.R == 6) return launch_segment_R<6>(M, V, T, P, I, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, st);
    if (V.R == 8) return launch_segment_R<8>(M, V, T, P, I, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, st);
    if (V.R == 12) return launch_segment_R<12>(M, V, T, P, I, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, st);
    return launch_segment_R<16>(M, V, T, P, I, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, st);
}

}  // namespace mcd
