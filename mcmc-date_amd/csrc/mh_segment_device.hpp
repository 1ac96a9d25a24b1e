// mh_segment_device.hpp -- what the two segment kernels share (k_mh_segment.hip: the factor of Sigma dense, z = L^-1 (d - mu) kept per
// chain; k_mh_segment_sparse.hip: the precision matrix sparse, the quadratic form updated through the rows of the moved distances):
// the hand-over words between a chain's two waves and the whole CHAIN WAVE -- state and kept prior summands in LDS, propose on the
// written nodes, post the transform, ln prior of the proposal, wait for the likelihood wave's |z'|^2 (here: q'), decide, commit or take
// back.  The likelihood waves differ and live with their kernels.  See k_mh_segment.hip for the story and the reference citations
// (`mhg`'s iteration, app/Main.hs:460-479; cycle app/Definitions.hs:256-278; likelihood app/Probability.hs:166-184, 195-207).
#pragma once
#include "mh_device.hpp"
#include "prior_device.hpp"

namespace mcd {

constexpr int kSegApplyDoubles = (int)((sizeof(PropApply) + 7) / 8);   // the proposal's per-node transform, chain wave -> likelihood wave
// The CLOCK prior wave also draws the NEXT step's proposal (MhSegPending::prior_draws): a step's two prior waves idle for most of it -- on
// a proposal that moves heights only the clock wave of an uncorrelated model has nothing to do at all -- while the chain wave would draw the
// next proposal only after the decision (or, round 4's first arrangement, ahead of it in its own idle time -- and then again after every
// acceptance).  The prior wave draws from the current state (what the next step starts from if this one is rejected: most are) with its
// own copy of the 64-step draw blocks and the scalars it keeps anyway; the chain wave takes the transform from here after a rejection
// and draws itself after an acceptance.  The same function on the same inputs: the same bits.
struct SegSpec {
    PropApply A;                   // the transform of step `word` - 1, its proposed scalars, ln (q-ratio * Jacobian)
    double sc1[5];
    double lnqj;
    int word, pad;
};
constexpr int kSegSpecDoubles = (int)((sizeof(SegSpec) + 7) / 8);
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
    return 6 * (size_t)n_nodes + (size_t)np + 2 * (size_t)kSegList + (size_t)kSegList / 2 + (size_t)np / 2 + 8 + 16 /* SegHelpWords */ + (size_t)kSegApplyDoubles +
           (size_t)kSegSpecDoubles;
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

// The hand-over between the chain wave and the two PRIOR waves of a chain (HELP = true): the chain wave posts the proposed scalars with
// the transform (SegWords::req), each prior wave answers with its block of the ln prior and reports when it has committed or taken back
// its summands after the decision (SegWords::dec).
struct SegHelpWords {
    int resp_bd, resp_cl;          // prior waves: step + 1 when c1p / c2p are there
    int done_bd, done_cl;          // ... when the block's summands are those of the state the decision left
    double c1p, c2p;               // the birth-death / the clock block of the ln prior of the proposal
    double sc1[5];                 // chain wave: the proposal's scalars
    int sel, pad_i;                // ... which buffers of MhDev::psum hold the current summands (the proposal after the segment: into the others)
    double pad[6];
};
static_assert(sizeof(SegHelpWords) == 128, "sixteen doubles of LDS");
constexpr int kSegHelpDoubles = 16;
constexpr int kSegAheadFrom = 200;                         // (MhSegPending::ahead_from; measured: profiles/r04_segment_ahead.txt)

// LDS of one chain's CHAIN wave and the constants of its likelihood: set up by the kernel, read by seg_chain_wave
struct SegChainCtx {
    SegHelpWords* help;                                             // (HELP = true)
    int32_t *tb_parent, *tb_size, *tb_first, *tb_nch, *tb_second;   // the tree's tables (LDS; tb_size may be the global table)
    double *Hc, *Rc, *Hp, *Rp, *tbd, *tcl;                          // [n_nodes] each: current / proposed state, summands of the two blocks
    SegWords* words;
    PropApply* A_lds;
    SegSpec* spec;                                                  // (HELP: the next step's proposal, drawn by the clock prior wave)
    double c, logdet;                                               // ll = c - 1/2 (logdet + q)
};

// One of a chain's two PRIOR waves (HELP = true), for the whole segment: BLOCK 0 the birth-death block, BLOCK 1 the clock block of the ln
// prior of every proposal -- the code the chain wave runs without them, on the same numbers: a proposal that writes a few nodes has the
// summands of those nodes re-evaluated in place (the old values wait in this wave's registers for the decision) and the block is the sum over
// kept and new summands in the order of the full evaluation; otherwise every summand.  Pl: the prior's tables with the tree in LDS.
template <int BLOCK>
__device__ __forceinline__ void seg_prior_wave(const MhDev& M, const PriorDev& P, const PriorDev& Pst, const SegChainCtx& L, const MhSegPending& Q,
                                               const int32_t* __restrict__ sched, int64_t n_steps, uint64_t step0, uint64_t seed, int64_t b, bool valid,
                                               int lane)
{
    PriorDev Pl = Pst;                                       // (as the chain wave: the tree's tables from LDS)
    Pl.parent = L.tb_parent;
    Pl.first_child = L.tb_first;
    Pl.n_children = L.tb_nch;
    Pl.second_child = L.tb_second;
    const int nn = M.n_nodes;
    const int64_t B = M.batch;
    int32_t* tb_first = L.tb_first;
    int32_t* tb_nch = L.tb_nch;
    int32_t* tb_second = L.tb_second;
    double* Hp = L.Hp;
    double* Rp = L.Rp;
    double* tbd = L.tbd;
    double* tcl = L.tcl;
    lds_vint_t* w_req = lds_vint(&L.words->req);
    lds_vint_t* w_dec = lds_vint(&L.words->dec);
    lds_vint_t* w_resp = lds_vint(BLOCK == 0 ? &L.help->resp_bd : &L.help->resp_cl);
    lds_vint_t* w_done = lds_vint(BLOCK == 0 ? &L.help->done_bd : &L.help->done_cl);
    lds_vdouble_t* w_val = lds_vdouble(BLOCK == 0 ? &L.help->c1p : &L.help->c2p);
    double la_pending;
    const bool took = Q.p_acc >= 0 && seg_accept_pending(M, Q, b, seed, la_pending);
    double sc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) sc[i] = (took ? M.sc1 : M.sc)[i * B + b];
    ClockCache cc{__builtin_nan(""), 0.0, 0.0, 0.0};
    if (BLOCK == 1) prior_clock_scalars(sc[4], cc);
    // (SegSpec) the clock wave draws the next step's proposal: the chain's tables, tuning parameters and draw blocks as the chain wave has them
    const bool draws = Q.prior_draws != 0;
    MhDev Ml = M;
    Ml.parent = L.tb_parent;
    Ml.size = L.tb_size;
    const double* tune = M.tune + b * M.n_prop;
    const int vz = mh_vzero();
    int p_n = 0, p_nn = 0;
    PropRow row_n{0, 0, 0, 0, 0, 0.0, 0.0};
    double tune_n = 0.0;
    StepDraws pre{1.0, 0.0, 0.0, 0.5, 0.5};
    int64_t blk = -1;
    if (draws) {
        p_n = sched[(n_steps > 1 ? 1 : 0) + vz];             // the row of step gs + 1 travels a step ahead, the schedule's entry after it two
        row_n = mh_load_row_ahead(M, p_n);
        tune_n = tune[p_n];
        p_nn = sched[(n_steps > 2 ? 2 : 0) + vz];
    }
    for (int64_t gs = 0; gs < n_steps; ++gs) {
        const int tag = (int)gs + 1;
        (void)seg_poll(w_req, tag, 0);
        const PropApply A = *L.A_lds;
        double sc1[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) sc1[i] = *lds_vdouble(&L.help->sc1[i]);
        const bool dH = A.hhi > A.hlo || A.hhi2 > A.hlo2 || A.pt1 >= 0 || A.pt2 >= 0 || A.brace_hi > A.brace_lo;
        const bool dR = A.rhi > A.rlo || A.rp1 >= 0 || A.rp2 >= 0 || A.rp3 >= 0 || (A.brace_hi > A.brace_lo && A.kind == MCD_PROP_SLIDE_BRACE_CONTRA);
        const int nbr = A.brace_hi - A.brace_lo;
        bool need = false, few = false, mine = false;
        int v = -1;
        double old = 0.0;
        ClockCache ccp = cc;
        if (BLOCK == 0) {
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
            const bool bd_scalars = sc1[0] != sc[0] || sc1[1] != sc[1];
            need = dH || bd_scalars;
            few = need && !bd_scalars && cnt_bd <= 64 && !prior_bd_near(sc1[0], sc1[1]);
            v = few ? cand_bd(lane) : -1;
            mine = few && lane < cnt_bd && v >= 1;
            if (few) {
                // in place (a node may come twice: every lane reads the old value before any lane writes -- LDS keeps a wave's order)
                if (mine) old = tbd[v];
                const double t = mine ? prior_bd_term(Pl, v, false, sc1[0], sc1[1], Hp) : 0.0;
                if (mine) tbd[v] = t;
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                double bd = 0.0;
                for (int w = 1 + lane; w < nn; w += 64) bd += tbd[w];
                const double c1p = prior_bd_finish(pr_wave_sum(bd), sc1[0], sc1[1]);
                if (lane == 0) *w_val = c1p;
            } else if (need) {
                const bool near = prior_bd_near(sc1[0], sc1[1]);
                double bd = 0.0;
                for (int w = 1 + lane; w < nn; w += 64) bd += prior_bd_term(Pl, w, near, sc1[0], sc1[1], Hp);
                const double c1p = prior_bd_finish(pr_wave_sum(bd), sc1[0], sc1[1]);
                if (lane == 0) *w_val = c1p;
            }
        } else {
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
            const bool cl_heights = dH && P.clock_model >= 2;    // white noise / autocorrelated: the summands also hold branch durations
            need = dR || sc1[3] != sc[3] || sc1[4] != sc[4] || cl_heights;
            few = need && sc1[4] == sc[4] && P.clock_model < 2 && cnt_cl <= 64;
            v = few ? cand_cl(lane) : -1;
            mine = few && lane < cnt_cl && v >= 1;
            if (few) {
                if (mine) old = tcl[v];
                const double t = mine ? prior_clock_term(Pl, v, sc1[4], cc.lg_k, cc.log_t, Hp, Rp) : 0.0;
                if (mine) tcl[v] = t;
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                double cl = 0.0;
                for (int w = 1 + lane; w < nn; w += 64) cl += tcl[w];
                const double c2p = prior_clock_finish(Pl, pr_wave_sum(cl), sc1[3], sc1[4], cc.hyper);
                if (lane == 0) *w_val = c2p;
            } else if (need) {
                if (ccp.va != sc1[4]) prior_clock_scalars(sc1[4], ccp);
                double cl = 0.0;
                for (int w = 1 + lane; w < nn; w += 64) cl += prior_clock_term(Pl, w, sc1[4], ccp.lg_k, ccp.log_t, Hp, Rp);
                const double c2p = prior_clock_finish(Pl, pr_wave_sum(cl), sc1[3], sc1[4], ccp.hyper);
                if (lane == 0) *w_val = c2p;
            }
        }
        if (need) seg_post(w_resp, tag);
        if (draws && gs + 1 < n_steps) {
            // ---- the proposal of step gs + 1 from the CURRENT state (read while the chain wave may already commit an accepted step gs: then
            // what is drawn here is not used).  The wave whose block this step does not need draws it -- the clock wave if both or neither do
            // (the same expressions on the same numbers in both waves).
            const bool n_bd = dH || sc1[0] != sc[0] || sc1[1] != sc[1];
            const bool n_cl = dR || sc1[3] != sc[3] || sc1[4] != sc[4] || (dH && P.clock_model >= 2);
            const int drawer = (n_cl && !n_bd) ? 0 : 1;
            const int64_t t = gs + 1;
            if (drawer == BLOCK) {
                if ((t >> 6) != blk) {
                    blk = t >> 6;
                    const int64_t mine = (blk << 6) + lane;
                    if (mine < n_steps) {
                        const int pl = sched[mine];
                        pre = mh_step_draws(mh_load_row(M, pl), tune[pl], mh_rng(seed, M.chain0 + b, step0 + (uint64_t)mine));
                    }
                }
                const int sl = (int)(t & 63);
                const StepDraws dr{mh_readlane64(pre.u, sl), mh_readlane64(pre.lnq, sl), mh_readlane64(pre.logu, sl), mh_readlane64(pre.U, sl), 0.5};
                double scn[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) scn[i] = sc[i];
                PropApply An;
                const double lqn = mh_propose_params(Ml, mh_row_scalar(row_n), tune_n, dr, lane, scn, L.Hc, L.Rc, An);
                if (lane == 0) {
                    L.spec->A = An;
#pragma unroll
                    for (int i = 0; i < 5; ++i) L.spec->sc1[i] = scn[i];
                    L.spec->lnqj = lqn;
                }
                seg_post(lds_vint(&L.spec->word), (int)t + 1);
            }
            p_n = p_nn;                                      // the row after it: on its way until the next step's draw
            row_n = mh_load_row_ahead(M, p_n);
            tune_n = tune[p_n];
            p_nn = sched[((t + 2 < n_steps) ? t + 2 : t) + vz];
        }
        const int d = seg_poll(w_dec, tag, 1);
        if (d & 1) {
            // (few: already in place)  every summand: evaluated again, now to be kept -- the same function results as the sum's
            if (!few && need) {
                if (BLOCK == 0) {
                    const bool near = prior_bd_near(sc1[0], sc1[1]);
                    for (int w = 1 + lane; w < nn; w += 64) tbd[w] = prior_bd_term(Pl, w, near, sc1[0], sc1[1], Hp);
                } else {
                    for (int w = 1 + lane; w < nn; w += 64) tcl[w] = prior_clock_term(Pl, w, sc1[4], ccp.lg_k, ccp.log_t, Hp, Rp);
                }
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) sc[i] = sc1[i];
            cc = ccp;
        } else if (mine) {
            if (BLOCK == 0) tbd[v] = old; else tcl[v] = old;     // the overwritten summand back
        }
        seg_post(w_done, tag);
    }
    // ---- the dense proposal after the segment (seg_chain_wave, MhSegPending::p_tail): this wave's block of its ln prior, every summand
    // into the buffer of MhDev::psum that is not the current one -- k_mh_step_wg's proposal half
    if (Q.p_tail >= 0) {
        const int tag = (int)n_steps + 1;
        (void)seg_poll(w_req, tag, 0);
        const PropApply A = *L.A_lds;
        double sc1[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) sc1[i] = *lds_vdouble(&L.help->sc1[i]);
        const int sel = *lds_vint(&L.help->sel);
        const size_t NS = (size_t)((nn - 1 + 63) / 64) * 64;
        const bool dH = A.hhi > A.hlo || A.hhi2 > A.hlo2 || A.pt1 >= 0 || A.pt2 >= 0 || A.brace_hi > A.brace_lo;
        const bool dR = A.rhi > A.rlo || A.rp1 >= 0 || A.rp2 >= 0 || A.rp3 >= 0 || (A.brace_hi > A.brace_lo && A.kind == MCD_PROP_SLIDE_BRACE_CONTRA);
        if (BLOCK == 0) {
            if (dH || sc1[0] != sc[0] || sc1[1] != sc[1]) {
                double* s_bd = M.psum + ((size_t)b * 4 + (size_t)((sel & 1) ^ 1)) * NS;
                const bool near = prior_bd_near(sc1[0], sc1[1]);
                double bd = 0.0;
                for (int w = 1 + lane; w < nn; w += 64) {
                    const double t = prior_bd_term(Pl, w, near, sc1[0], sc1[1], Hp);
                    if (valid) s_bd[w - 1] = t;
                    bd += t;
                }
                const double c1p = prior_bd_finish(pr_wave_sum(bd), sc1[0], sc1[1]);
                if (lane == 0) *w_val = c1p;
                seg_post(w_resp, tag);
            }
        } else {
            if (dR || sc1[3] != sc[3] || sc1[4] != sc[4] || (dH && P.clock_model >= 2)) {
                double* s_cl = M.psum + ((size_t)b * 4 + 2 + (size_t)(((sel >> 1) & 1) ^ 1)) * NS;
                if (!(cc.va == sc1[4])) prior_clock_scalars(sc1[4], cc);
                double cl = 0.0;
                for (int w = 1 + lane; w < nn; w += 64) {
                    const double t = prior_clock_term(Pl, w, sc1[4], cc.lg_k, cc.log_t, Hp, Rp);
                    if (valid) s_cl[w - 1] = t;
                    cl += t;
                }
                const double c2p = prior_clock_finish(Pl, pr_wave_sum(cl), sc1[3], sc1[4], cc.hyper);
                if (lane == 0) *w_val = c2p;
                seg_post(w_resp, tag);
            }
        }
    }
}

// The likelihood wave's share of the dense proposal that follows the segment (MhSegPending::p_tail; seg_chain_wave proposes it): every
// distance of the proposed state and ln jacobianRootBranch -- the arithmetic of k_mh_step_wg's X1 (likelihoodFunctionWrapper,
// app/Probability.hs:195-207, 393-410).
template <typename SlotT>
__device__ __forceinline__ void seg_tail_distances(const MhDev& M, const MhSegPending& Q, SegWords* words, const double* Hp, const double* Rp, const SlotT* ts_node,
                                                   const SlotT* ts_parent, int rr, int n, int64_t n_steps, int64_t b, bool valid, int lane)
{
    if (Q.p_tail < 0) return;
    (void)seg_poll(lds_vint(&words->req), (int)n_steps + 1, 0);
    const double s1 = *lds_vdouble(&words->s1);
    double* x1 = Q.X1_tail + b * (int64_t)n;
    for (int j = lane; j < n; j += 64) {
        const int a = ts_node[j], pa = ts_parent[j];
        double d = (Hp[pa] - Hp[a]) * Rp[a];
        if (j == 0) d = d + (Hp[0] - Hp[rr]) * Rp[rr];
        d = d * s1;
        if (valid) {
            x1[j] = d;
            if (j == 0) M.post1[2 * M.batch + b] = log(1.0 / d);
        }
    }
}

// The chain wave of one chain for the whole segment (one wave; `lane` = its lane, b = the chain, valid = whether it exists).
// HELP: two more waves of the workgroup evaluate the birth-death and the clock block of the proposal's ln prior (seg_prior_wave below: the
// same functions on the same numbers in the same order, hence the same bits) while this wave evaluates the node priors -- the three
// blocks depend on the proposal only, and one wave evaluated them one after the other: 37 - 44 % of a step (profiles/r04_*_phases*).
template <bool HELP, bool PLAIN>
__device__ __forceinline__ void seg_chain_wave(const MhDev& M, const PriorDev& P, const PriorDev& Pst, const SegChainCtx& L, const MhSegPending& Q,
                                               const int32_t* __restrict__ sched, int64_t n_steps, int32_t S, int accumulate, uint64_t step0,
                                               uint64_t seed, double* __restrict__ trace_alpha, int8_t* __restrict__ trace_accept, int64_t gs_base,
                                               int summands_kept, int64_t b, bool valid, int lane)
{
    const int nn = M.n_nodes, NPr = M.n_prop;
    const int64_t B = M.batch;
    int32_t* tb_parent = L.tb_parent;
    int32_t* tb_size = L.tb_size;
    int32_t* tb_first = L.tb_first;
    int32_t* tb_nch = L.tb_nch;
    int32_t* tb_second = L.tb_second;
    double* Hc = L.Hc;
    double* Rc = L.Rc;
    double* Hp = L.Hp;
    double* Rp = L.Rp;
    double* tbd = L.tbd;
    double* tcl = L.tcl;
    SegWords* words = L.words;
    PropApply* A_lds = L.A_lds;
    lds_vint_t* w_req = lds_vint(&words->req);
    lds_vint_t* w_moves = lds_vint(&words->moves);
    lds_vint_t* w_resp = lds_vint(&words->resp);
    lds_vint_t* w_dec = lds_vint(&words->dec);
    lds_vint_t* w_cnt = lds_vint(&words->cnt);
    lds_vint_t* w_have0 = lds_vint(&words->have0);
    lds_vdouble_t* w_q = lds_vdouble(&words->q);
    lds_vdouble_t* w_lj = lds_vdouble(&words->lj);
    lds_vdouble_t* w_s1 = lds_vdouble(&words->s1);
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
    // ---- The steps.  A step is a dependent chain -- draw the proposal (special functions of one value per chain), then the ln prior and
    // the likelihood of the proposed state (the other waves), then the decision -- and this wave idles while the other waves evaluate.  So
    // it draws the NEXT step's proposal in that time, from the current state: that is the state the next step starts from if this step is
    // rejected (most are).  After an acceptance the proposal is drawn again from the new state.  One site of mh_propose_params in the loop:
    //   target = the step that is proposed in this turn of the loop: the step after the one in flight, or -- nothing in flight (the first
    //   turn, and after an acceptance) -- the step gs itself.
    // The same functions on the same inputs as a loop that proposes, evaluates and decides one step after the other: the same bits.
    const int vz = mh_vzero();                               // (mh_device.hpp: what travels ahead is loaded by vector loads)
    int p_t = sched[0];                                      // the target's row of the proposal table, its tuning parameter
    PropRow row_t = mh_load_row(M, p_t);
    double tune_t = tune[p_t];
    int64_t t_idx = 0;                                       // ... and which step that is
    int p_n = sched[(n_steps > 1 ? 1 : 0) + vz];             // the step after the target: its row travels a turn ahead
    PropRow row_n = mh_load_row_ahead(M, p_n);
    double tune_n = tune[p_n];
    int p_nn = sched[(n_steps > 2 ? 2 : 0) + vz];            // ... and the schedule's entry after that one (a row's loads need its index: a load of the
                                                             // index at the point of use made every step wait for two loads in a row)
    StepDraws pre{1.0, 0.0, 0.0, 0.5, 0.5};                  // lane l: the state-independent draws of step 64 blk + l
    int64_t blk = -1;
    // the step in flight: applied to Hp / Rp, posted to the other waves
    bool inflight = false;
    PropApply A;
    double sc1[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    double lnqj = 0.0, uacc = 0.0;
    int p = p_t, jac_root = 0;
    bool moves = false;
    int64_t gs = 0;
    // (small trees: the other waves answer before a proposal is drawn -- drawing ahead would only add the discarded draws after acceptances)
    const bool dual = !PLAIN && HELP && Q.prior_draws != 0;  // (SegSpec) the next proposal drawn for both outcomes of the decision
    const bool ahead = !PLAIN && (nn >= Q.ahead_from || dual);
    int64_t to_close = S - (gs_base % S);                    // steps until the next iteration of the cycle closes
    // a drawn proposal (step gs) goes into flight: applied to Hp / Rp, posted to the other waves
    auto launch = [&](const PropApply& An, const double (&sc1n)[5], double lnqjn, double uaccn) __attribute__((always_inline)) {
        const int tag = (int)gs + 1;
        A = An;
        mh_apply_scalars(A);
#pragma unroll
        for (int i = 0; i < 5; ++i) sc1[i] = sc1n[i];
        lnqj = lnqjn;
        uacc = uaccn;
        p = p_t;
        jac_root = row_t.jac_root;
        moves = seg_moves_likelihood(row_t.kind, row_t.node);
        if constexpr (HELP) {
            // (the prior waves read Hp / Rp until they have committed or taken back the previous step's summands)
            if (gs > 0) {
                (void)seg_poll(lds_vint(&L.help->done_bd), tag - 1, 0);
                (void)seg_poll(lds_vint(&L.help->done_cl), tag - 1, 0);
            }
        }
        SEG_TICK(6)
        for_write_set(A, [&](int w) {
            double h, r;
            mh_propose_node(Ml, A, w, Hc, Rc, h, r);
            Hp[w] = h;
            Rp[w] = r;
        });
        __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): the writes above have landed before any lane reads them
        __builtin_amdgcn_wave_barrier();
        SEG_TICK(7)
        // ---- the likelihood wave takes it from here: the transform (which nodes are written), Hp / Rp, tH * rMu
        if (lane == 0) {
            *A_lds = A;
            *w_s1 = sc1[2] * sc1[3];
            *w_moves = moves ? 1 : 0;
            if constexpr (HELP) {
#pragma unroll
                for (int i = 0; i < 5; ++i) L.help->sc1[i] = sc1[i];
            }
        }
        seg_post(w_req, tag);                            // (every lane stores the same word: the fence is the wave's)
        inflight = true;
        SEG_TICK(2)
    };
    // (PLAIN -- the sparse kernel's second instance of this function, for small trees: nothing is ever in flight at the loop's head and the
    // second site of `launch` goes away: the plain propose / evaluate / decide, 5 % faster at 13 nodes than this loop with `ahead` false)
    constexpr bool AHEAD = !PLAIN;
    while (gs < n_steps) {
        const int64_t target = (AHEAD && inflight) ? gs + 1 : gs;
        const bool do_prop = target < n_steps && (!inflight || ahead);
        const bool drew_next = AHEAD && inflight && do_prop; // An is the proposal of the step AFTER the one in flight
        if (do_prop && target != t_idx) {
            p_t = __builtin_amdgcn_readfirstlane(p_n);
            row_t = mh_row_scalar(row_n);
            tune_t = tune_n;
            t_idx = target;
            p_n = p_nn;
            row_n = mh_load_row_ahead(M, p_n);               // (global memory: a load at the point of use stalled the proposal)
            tune_n = tune[p_n];
            p_nn = sched[((target + 2 < n_steps) ? target + 2 : target) + vz];
        }
        PropApply An;
        double sc1n[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) sc1n[i] = sc[i];
        double lnqjn = 0.0, uaccn = 0.0;
        SEG_TICK(0)
        if (do_prop) {
            if ((target >> 6) != blk) {
                // 64 consecutive steps at once, one step per lane: what can be drawn knowing only the proposal row and its tuning
                // parameter (as k_mh_draws does for the two-launch path)
                blk = target >> 6;
                const int64_t mine = (blk << 6) + lane;
                if (mine < n_steps) {
                    const int pl = sched[mine];
                    pre = mh_step_draws(mh_load_row(M, pl), tune[pl], mh_rng(seed, M.chain0 + b, step0 + (uint64_t)mine));
                }
            }
            const int sl = (int)(target & 63);
            const StepDraws dr{mh_readlane64(pre.u, sl), mh_readlane64(pre.lnq, sl), mh_readlane64(pre.logu, sl), mh_readlane64(pre.U, sl),
                               mh_readlane64(pre.Uacc, sl)};
            // (SegSpec) with a prior wave drawing the next proposal from the CURRENT state -- the branch "this step is rejected" --, this wave
            // draws it from the PROPOSED state: the branch "accepted".  Whatever the decision, the next proposal is there.
            const bool from_proposed = dual && inflight;
            if (from_proposed) {
#pragma unroll
                for (int i = 0; i < 5; ++i) sc1n[i] = sc1[i];
            }
            lnqjn = mh_propose_params(Ml, row_t, tune_t, dr, lane, sc1n, from_proposed ? Hp : Hc, from_proposed ? Rp : Rc, An);
            uaccn = dr.Uacc;
        }
        SEG_TICK(1)
        if (!inflight) {                                     // An proposes step gs itself
            launch(An, sc1n, lnqjn, uaccn);
            if (AHEAD && ahead) continue;                    // (the next turn draws the step after it, then takes the answers)
        }
        bool start = false;
        {
            // ================ the step in flight: ln prior, the likelihood wave's answer, the decision
            const int tag = (int)gs + 1;
            bool bd_scalars = false, need_bd = false, few_bd = false, need_cl = false, few_cl = false, cl_heights = false, mine_bd = false, mine_cl = false;
            int v_bd = -1, v_cl = -1;
            double old_bd = 0.0, old_cl = 0.0;               // the summands this lane overwrites, until the decision
            double c1p = c1, c2p = c2;
            double lj1 = lj;
            ClockCache ccp = cc;
            // ---- ln prior: only the blocks whose inputs the proposal writes (a superset of "changed": a block re-evaluated on unchanged
            // inputs returns the same bits)
            const bool dH = A.hhi > A.hlo || A.hhi2 > A.hlo2 || A.pt1 >= 0 || A.pt2 >= 0 || A.brace_hi > A.brace_lo;
            const bool dR = A.rhi > A.rlo || A.rp1 >= 0 || A.rp2 >= 0 || A.rp3 >= 0 || (A.brace_hi > A.brace_lo && A.kind == MCD_PROP_SLIDE_BRACE_CONTRA);
            ccp = cc;                                        // refreshed only if the proposal moved rVar
            const double c0p = (dH || sc1[2] != sc[2]) ? prior_nodes_wave(Pl, lane, sc1[2], Hp) : c0;
            if constexpr (!HELP) {
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
                bd_scalars = sc1[0] != sc[0] || sc1[1] != sc[1];
                need_bd = dH || bd_scalars;
                few_bd = need_bd && !bd_scalars && cnt_bd <= 64 && !prior_bd_near(sc1[0], sc1[1]);
                c1p = c1;
                v_bd = few_bd ? cand_bd(lane) : -1;
                mine_bd = few_bd && lane < cnt_bd && v_bd >= 1;
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
                cl_heights = dH && P.clock_model >= 2;       // white noise / autocorrelated: the summands also hold branch durations
                need_cl = dR || sc1[3] != sc[3] || sc1[4] != sc[4] || cl_heights;
                few_cl = need_cl && sc1[4] == sc[4] && P.clock_model < 2 && cnt_cl <= 64;
                c2p = c2;
                v_cl = few_cl ? cand_cl(lane) : -1;
                mine_cl = few_cl && lane < cnt_cl && v_cl >= 1;
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
            } else {
                // the two prior waves evaluate their blocks (seg_prior_wave); whether a block has to be evaluated at all is decided here and
                // there alike (the same expressions on the same numbers)
                need_bd = dH || sc1[0] != sc[0] || sc1[1] != sc[1];
                need_cl = dR || sc1[3] != sc[3] || sc1[4] != sc[4] || (dH && P.clock_model >= 2);
                if (need_bd) {
                    (void)seg_poll(lds_vint(&L.help->resp_bd), tag, 0);
                    c1p = *lds_vdouble(&L.help->c1p);
                }
                if (need_cl) {
                    (void)seg_poll(lds_vint(&L.help->resp_cl), tag, 0);
                    c2p = *lds_vdouble(&L.help->c2p);
                }
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
                if (cnt < 0) ll = __builtin_nan("");         // more moved distances than the list holds -- cannot happen for a proposal mh_capi.cpp
                                                             // put into a segment; if it does, the chain says so: its ln likelihood is NaN from here
                                                             // on (nothing is accepted, mcd_mh_get_posterior shows it), not a silently wrong value
                ll1 = (cnt < 0) ? __builtin_nan("") : L.c + (-0.5) * (L.logdet + q);      // :169 (finish_ll)
            }
            double la = beta * ((lp1 + ll1) - (lp + ll)) + lnqj;           // heated chains of MC3: posterior^beta; beta = 1 is exact
            if (jac_root) la += (double)jac_root * (lj1 - lj);
            // (every lane holds the same numbers: the decision as a scalar, so that the loop's control flow -- what is in flight, whether the
            // proposal drawn ahead is used -- stays the wave's, not per lane)
            const bool ok = __builtin_amdgcn_readfirstlane(((la >= 0) || (uacc < exp(la))) ? 1 : 0) != 0;
            seg_post(w_dec, 2 * tag + (ok ? 1 : 0));
            if (ok) {
                for_write_set(A, [&](int w) {
                    Hc[w] = Hp[w];
                    Rc[w] = Rp[w];
                });
#pragma unroll
                for (int i = 0; i < 5; ++i) sc[i] = sc1[i];
                // (few: already in place)  every summand: evaluated again, now to be kept -- the same function results as the sum's
                if constexpr (!HELP) {
                    if (!few_bd && need_bd) {
                        const bool near = prior_bd_near(sc1[0], sc1[1]);
                        for (int v = 1 + lane; v < nn; v += 64) tbd[v] = prior_bd_term(Pl, v, near, sc1[0], sc1[1], Hp);
                    }
                    if (!few_cl && need_cl)
                        for (int v = 1 + lane; v < nn; v += 64) tcl[v] = prior_clock_term(Pl, v, sc1[4], ccp.lg_k, ccp.log_t, Hp, Rp);
                }
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
                if constexpr (!HELP) {
                    if (mine_bd) tbd[v_bd] = old_bd;         // the overwritten summands back
                    if (mine_cl) tcl[v_cl] = old_cl;
                }
            }
            if (lane == 0 && valid) {
                atomicAdd(&tried[p], 1);                     // (global memory, no value returned: nothing waits for it)
                if (ok) atomicAdd(&acc[p], 1);
                if (trace_alpha) trace_alpha[gs * B + b] = la;
                if (trace_accept) trace_accept[gs * B + b] = ok ? 1 : 0;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            to_close -= 1;
            if (to_close == 0) {                             // once per iteration of the cycle: straight into the running sums
                to_close = S;
                if (accumulate && valid) {
                    for (int w = lane; w < nn; w += 64) {
                        const double a = sc[2] * Hc[w];
                        M.age_sum[b * nn + w] += a;
                        M.age_sq[b * nn + w] += a * a;
                    }
                }
            }
            SEG_TICK(5)
            gs += 1;
            inflight = false;
            if (gs >= n_steps) break;
            if (dual) {
                start = drew_next;                           // accepted: An was drawn from the state that is now the current one
                if (!ok && drew_next) {                      // rejected: the prior wave's draw from the state that still is
                    (void)seg_poll(lds_vint(&L.spec->word), (int)gs + 1, 0);
                    An = L.spec->A;
#pragma unroll
                    for (int i = 0; i < 5; ++i) sc1n[i] = L.spec->sc1[i];
                    lnqjn = L.spec->lnqj;
                }
            } else {
                start = !ok && drew_next;                    // rejected: the state An was drawn from is still the current one
            }
        }
        if constexpr (AHEAD) {
            if (start) launch(An, sc1n, lnqjn, uaccn);
        }
    }
#ifdef MCD_SEG_STAMP
    if (trace_alpha && lane == 0 && valid)
        for (int i = 0; i < 8; ++i) trace_alpha[(int64_t)i * B + b] = (double)tk[i];
#endif
    if constexpr (HELP) {
        if (n_steps > 0) {
            (void)seg_poll(lds_vint(&L.help->done_bd), (int)n_steps, 0);
            (void)seg_poll(lds_vint(&L.help->done_cl), (int)n_steps, 0);
        }
    }
    // ---- the dense proposal that follows the segment (MhSegPending::p_tail): PROPOSED here, from the state this wave holds in LDS --
    // k_mh_step_wg's proposal half (k_mh.hip) on the same numbers with the same functions in the same order: H1 / R1 / sc1, the three blocks
    // of its ln prior (pcomp1, post1[0]), lnqj, its summands in the buffers that are not the current ones, psel's flags; the likelihood
    // wave writes its distances (seg_tail_distances).  The row-split (or sparse) launch evaluates it, the next launch decides it.
    int tail_flags = 0;
    if (Q.p_tail >= 0) {
        const int pt = Q.p_tail;
        const PropRow row_t = mh_load_row(M, pt);
        const StepDraws dr = mh_step_draws(row_t, tune[pt], mh_rng(seed, M.chain0 + b, step0 + (uint64_t)n_steps));     // (every lane the same)
        double sc1[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) sc1[i] = sc[i];
        PropApply A;
        const double lnqj = mh_propose_params(Ml, row_t, tune[pt], dr, lane, sc1, Hc, Rc, A);
        for_write_set(A, [&](int w) {
            double h, r;
            mh_propose_node(Ml, A, w, Hc, Rc, h, r);
            Hp[w] = h;
            Rp[w] = r;
        });
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            *w_s1 = sc1[2] * sc1[3];
            if constexpr (HELP) {
                *A_lds = A;
#pragma unroll
                for (int i = 0; i < 5; ++i) L.help->sc1[i] = sc1[i];
                L.help->sel = seg_sel;
            }
        }
        seg_post(w_req, (int)n_steps + 1);                   // the likelihood wave (and the prior waves): the proposed state is in Hp / Rp
        // which blocks the proposal writes (a superset of k_mh_step_wg's "differs": a block evaluated again on unchanged inputs is the same
        // bits, and psel's flags say which buffers hold the proposal's summands either way)
        const bool dH = A.hhi > A.hlo || A.hhi2 > A.hlo2 || A.pt1 >= 0 || A.pt2 >= 0 || A.brace_hi > A.brace_lo;
        const bool dR = A.rhi > A.rlo || A.rp1 >= 0 || A.rp2 >= 0 || A.rp3 >= 0 || (A.brace_hi > A.brace_lo && A.kind == MCD_PROP_SLIDE_BRACE_CONTRA);
        const bool f_nodes = dH || sc1[2] != sc[2];
        const bool f_bd = dH || sc1[0] != sc[0] || sc1[1] != sc[1];
        const bool f_cl = dR || sc1[3] != sc[3] || sc1[4] != sc[4] || (dH && P.clock_model >= 2);
        tail_flags = (f_nodes ? 1 : 0) | (f_bd ? 2 : 0) | (f_cl ? 4 : 0);
        const double c0p = f_nodes ? prior_nodes_wave(Pl, lane, sc1[2], Hp) : c0;
        double c1p = c1, c2p = c2;
        if constexpr (HELP) {                                // (seg_prior_wave: the same expressions decide there whether a block is evaluated)
            if (f_bd) {
                (void)seg_poll(lds_vint(&L.help->resp_bd), (int)n_steps + 1, 0);
                c1p = *lds_vdouble(&L.help->c1p);
            }
            if (f_cl) {
                (void)seg_poll(lds_vint(&L.help->resp_cl), (int)n_steps + 1, 0);
                c2p = *lds_vdouble(&L.help->c2p);
            }
        } else {
        if (f_bd) {
            double* s_bd = M.psum + ((size_t)b * 4 + (size_t)((seg_sel & 1) ^ 1)) * NS;
            const bool near = prior_bd_near(sc1[0], sc1[1]);
            double bd = 0.0;
            for (int v = 1 + lane; v < nn; v += 64) {
                const double t = prior_bd_term(Pl, v, near, sc1[0], sc1[1], Hp);
                if (valid) s_bd[v - 1] = t;
                bd += t;
            }
            c1p = prior_bd_finish(pr_wave_sum(bd), sc1[0], sc1[1]);
        }
        if (f_cl) {
            double* s_cl = M.psum + ((size_t)b * 4 + 2 + (size_t)(((seg_sel >> 1) & 1) ^ 1)) * NS;
            ClockCache ccp = cc;
            if (!(ccp.va == sc1[4])) prior_clock_scalars(sc1[4], ccp);
            double cl = 0.0;
            for (int v = 1 + lane; v < nn; v += 64) {
                const double t = prior_clock_term(Pl, v, sc1[4], ccp.lg_k, ccp.log_t, Hp, Rp);
                if (valid) s_cl[v - 1] = t;
                cl += t;
            }
            c2p = prior_clock_finish(Pl, pr_wave_sum(cl), sc1[3], sc1[4], ccp.hyper);
        }
        }
        if (valid) {
            for (int w = lane; w < nn; w += 64) {
                M.H1[b * M.ld + w] = Hp[w];
                M.R1[b * M.ld + w] = Rp[w];
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 5; ++i) M.sc1[i * B + b] = sc1[i];
                M.pcomp1[b * 3 + 0] = c0p;
                M.pcomp1[b * 3 + 1] = c1p;
                M.pcomp1[b * 3 + 2] = c2p;
                M.lnqj[b] = lnqj;
                M.post1[b] = c0p + c1p + c2p;
            }
        }
    }
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
        if (lane == 0) reinterpret_cast<int2*>(M.psel)[b] = make_int2(seg_sel, tail_flags);
    }
}

}  // namespace mcd
