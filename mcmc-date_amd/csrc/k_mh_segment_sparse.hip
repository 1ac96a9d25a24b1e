// k_mh_segment_sparse.hip -- Metropolis-Hastings-Green over a SPARSE precision matrix (gfx950): a segment of a schedule in one launch,
// the chains' states in LDS -- the reference's production configuration (`mhg` with likelihoodFunction (Sparse ...), app/Main.hs:474,
// 257-277, 333-347; logDensitySparseMultivariateNormal, app/Probability.hs:178-184; every published timing of the reference uses it).
// SURVEY.md 8(f) row f2 over 8(a) A6.
//
// Round 3 ran this configuration with four launches per lock step and a full product P dx for every proposal (94 us per lock step at
// 2013 nodes x 512 chains).  With a sparse P the INCREMENTAL form is the cheap one: a proposal moves the distances J (one to three for
// most of the cycle), and with dx = d - mu, delta = d' - d,
//     q' = (dx + delta)^T P (dx + delta) = q + sum_{j in J} delta_j  sum_k Ps[j][k] (2 dx_k + delta_k),     Ps = (P + P^T) / 2,
// touches only the ROWS j in J of the matrix (about 15 nonzeros each in a graphical-lasso estimate) -- no column of a dense L^-1, nothing
// kept per chain but q and the current distances.  The kernel is k_mh_segment.hip's with another likelihood wave:
//   chain wave       shared code (mh_segment_device.hpp): state and kept prior summands in LDS, propose on the written nodes, ln prior of
//                    the proposal, decision, commit / take back -- the same bits as every other path of the driver;
//   likelihood wave  lists the distance slots the written nodes feed (each once: an LDS exchange per candidate), their new distances
//                    from the proposed state in LDS and the deltas against the current distances (global memory, MhInc::X0: one batch of
//                    loads); then lanes = listed slots: a lane walks its row of Ps, 16 entries in flight, (2 dx_k + delta_k) from X0, mu and
//                    the list (an LDS look-up through the slots' marks); q' = q + the lanes' sum in a fixed order; on accept q and the
//                    listed distances are committed.
// One chain per workgroup above about 1100 nodes (a chain's six state rows fill the LDS: 2013 nodes = 159 KiB), two below.
// Proposals that move more distances than the list holds (kSsegList) -- scalings of the whole tree, large sub trees: about 5 % of the
// cycle -- take two launches: k_mh_step_wg proposes, k_sparse_quad (k_sparse.hip) evaluates the full form, the next segment decides.
// Every 256 steps q is recomputed by a full product (mh_capi.cpp).  Parity: decisions, states, ln priors and ln Jacobians the same bits
// as the four-launch path; ln likelihoods agree to rounding (tests/test_gpu_mh.py::test_metropolis_hastings_over_a_sparse_likelihood).
#include "mvn_device.hpp"
#include "mh_segment_device.hpp"
#include "options.h"

#include <atomic>

namespace mcd {

constexpr int kSsegList = 254;           // moved distances of one proposal at most: a slot's mark holds its list position in 8 bits
constexpr int kSsegListAlloc = 256;

// LDS, in doubles.  The tree's tables: parent, first / second child, number of children (int32 [n]), the sub tree sizes where they still fit
// (else read from global memory: one scalar load per proposal); slot -> node, slot -> parent (int16 [np]), node -> slot (int16 [n]).
__host__ __device__ inline size_t sseg_table_doubles(int n_nodes, int np, int size_in_lds)
{
    return ((4 + (size_t)size_in_lds) * (size_t)n_nodes + 1) / 2 + 1 + ((size_t)n_nodes + 2 * (size_t)np + 3) / 4 + 1;
}
// Per chain: 4 state rows and the summands of the two blocks [n]; the list (new distance, delta: doubles; slot: int32); the slots' marks
// (int32 [np]); eight words of hand-over; the proposal's per-node transform.
__host__ __device__ inline size_t sseg_chain_doubles(int n_nodes, int np)
{
    return 6 * (size_t)n_nodes + 2 * (size_t)kSsegListAlloc + (size_t)kSsegListAlloc / 2 + (size_t)np / 2 + 8 + (size_t)kSegHelpDoubles + (size_t)kSegApplyDoubles +
           (size_t)kSegSpecDoubles;
}
__host__ __device__ inline size_t sseg_lds_bytes(int n_nodes, int np, int cpw, int size_in_lds)
{
    return sizeof(double) * (sseg_table_doubles(n_nodes, np, size_in_lds) + (size_t)cpw * sseg_chain_doubles(n_nodes, np));
}
constexpr size_t kSsegLdsMax = 160 * 1024;
// chains per workgroup and whether the sub tree sizes live in LDS (0 chains: the tree does not fit at all)
static inline void sseg_geometry(int n_nodes, int np, int& cpw, int& size_in_lds)
{
    for (int c = 2; c >= 1; --c)
        for (int s = 1; s >= 0; --s)
            if (sseg_lds_bytes(n_nodes, np, c, s) <= kSsegLdsMax) {
                cpw = c;
                size_in_lds = s;
                return;
            }
    cpw = 0;
    size_in_lds = 0;
}
__host__ __device__ inline size_t sseg_node_tables_bytes(int n_nodes, int np, int cpw, int size_in_lds, int n_cal, int n_con)
{
    const size_t need = sizeof(double) * prior_node_tables_doubles(n_cal, n_con);
    return (need > 0 && sseg_lds_bytes(n_nodes, np, cpw, size_in_lds) + need <= kSsegLdsMax) ? need : 0;
}

// HELP: two more waves per chain evaluate the birth-death and the clock block of the ln prior beside the chain wave (mh_segment_device.hpp)
template <int CPW, bool HELP>
__global__ __launch_bounds__(64 * CPW * (HELP ? 4 : 2), 1) void k_mh_segment_sparse(MhDev M, SparseDev Sp, TreeDev T, PriorDev P, MhInc I, const int32_t* __restrict__ sched,
                                                                   int64_t n_steps, int32_t S, int accumulate, uint64_t step0, uint64_t seed,
                                                                   double* __restrict__ trace_alpha, int8_t* __restrict__ trace_accept, int64_t gs_base,
                                                                   int summands_kept, MhSegPending Q, int size_in_lds, int list_all)
{
    extern __shared__ double dyn[];
    constexpr int NT = 64 * CPW * (HELP ? 4 : 2);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cs = wave % CPW;                               // which of the workgroup's chains
    const int role = wave / CPW;                             // 0 the chain wave, 1 the likelihood wave, 2 / 3 the prior waves (HELP)
    const bool chain_role = role == 0;
    const int nn = M.n_nodes;
    const int n = Sp.n;                                      // distance slots (= nn - 2)
    const int NPad = (n + 63) / 64 * 64;
    const int64_t B = M.batch;
    const int64_t b_raw = (int64_t)blockIdx.x * CPW + cs;
    const bool valid = b_raw < B;                            // a chain beyond the batch works on the last chain's inputs and stores nothing
    const int64_t b = valid ? b_raw : B - 1;
    // ---- LDS
    int32_t* tb_parent = reinterpret_cast<int32_t*>(dyn);
    int32_t* tb_first = tb_parent + nn;
    int32_t* tb_nch = tb_first + nn;
    int32_t* tb_second = tb_nch + nn;
    int32_t* tb_size = size_in_lds ? tb_second + nn : const_cast<int32_t*>(M.size);
    int16_t* ts_node = reinterpret_cast<int16_t*>(dyn + ((4 + (size_t)size_in_lds) * (size_t)nn + 1) / 2 + 1);   // [NPad] slot -> node (-1 padded)
    int16_t* ts_parent = ts_node + NPad;                                                                          // [NPad] slot -> that node's parent
    int16_t* ts_of = ts_parent + NPad;                                                                            // [nn] node -> slot (-1: the root)
    double* chain0 = dyn + sseg_table_doubles(nn, NPad, size_in_lds) + (size_t)cs * sseg_chain_doubles(nn, NPad);
    double* Hc = chain0;
    double* Rc = Hc + nn;
    double* Hp = Rc + nn;
    double* Rp = Hp + nn;
    double* tbd = Rp + nn;
    double* tcl = tbd + nn;
    double* l_dnew = tcl + nn;                               // [kSsegListAlloc] the list of this step: new distance, delta, slot
    double* l_delta = l_dnew + kSsegListAlloc;
    int32_t* l_j = reinterpret_cast<int32_t*>(l_delta + kSsegListAlloc);
    int32_t* mark = l_j + kSsegListAlloc;                    // [NPad] (step + 1) << 8 | position in that step's list
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

    // ---- the tables, by all waves (the only workgroup barriers of the kernel: before any wave polls a hand-over word)
    const int rr = T.root_right;
    for (int v = threadIdx.x; v < nn; v += NT) {
        tb_parent[v] = M.parent[v];
        if (size_in_lds) tb_size[v] = M.size[v];
        tb_first[v] = P.first_child[v];
        tb_nch[v] = P.n_children[v];
        tb_second[v] = P.second_child[v];
        ts_of[v] = -1;
    }
    if (chain_role && lane == 0) {
        words->req = 0;
        words->moves = 0;
        words->resp = 0;
        words->dec = 0;
        words->cnt = 0;
        words->have0 = 0;
        spec->word = 0;
        help->resp_bd = 0;
        help->resp_cl = 0;
        help->done_bd = 0;
        help->done_cl = 0;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < NPad; j += NT) {
        const int a = (j < n) ? T.slot_node[j] : -1;         // (the sparse tree's tables are [n], not padded)
        ts_node[j] = (int16_t)a;
        ts_parent[j] = (int16_t)((j < n) ? T.slot_parent[j] : 0);
        if (a >= 0) ts_of[a] = (int16_t)j;
    }
    __syncthreads();
    if (threadIdx.x == 0) ts_of[rr] = 0;                     // the root's two daughters share slot 0 (sumFirstTwo); no slot feeds on rr itself
    PriorDev Pst = P;                                        // (the node priors' tables in LDS where they fit)
    if (sseg_node_tables_bytes(nn, NPad, CPW, size_in_lds, P.n_cal, P.n_con) > 0)
        prior_stage_node_tables(Pst, P, dyn + sseg_table_doubles(nn, NPad, size_in_lds) + (size_t)CPW * sseg_chain_doubles(nn, NPad), (int)threadIdx.x, NT);
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
    L.c = Sp.c;
    L.logdet = Sp.logdet;
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
        double la_pending;
        const bool took = Q.p_acc >= 0 && seg_accept_pending(M, Q, b, seed, la_pending);
        double* X0 = I.X0 + b * (int64_t)n;                  // the current distances of this chain (global memory; this wave alone reads and writes them)
        double q = took ? I.zprop[b] : I.zcur[b];            // the quadratic form dx^T P dx of the current state (MhInc of the sparse driver: NPz = 1)
        if (took && valid) {                                 // the accepted dense proposal's distances become the current ones
            const double* x1 = Q.X1 + b * (int64_t)n;
            for (int j = lane; j < n; j += 64) X0[j] = x1[j];
        }
        double s_cur = (took ? M.sc1 : M.sc)[2 * B + b] * (took ? M.sc1 : M.sc)[3 * B + b];   // tH * rMu of the current state
        for (int j = lane; j < NPad; j += 64) mark[j] = 0;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#ifdef MCD_SEG_STAMP
        // diagnostic build: ticks per phase of the likelihood wave in rows 8 .. 12 of trace_alpha: 8 waiting for the request, 9 ahead of it
        // (the list from the table row, the fetches issued), 10 request -> deltas, look-ups, sum, 11 the answer posted, 12 waiting for the
        // decision + commit
        uint64_t lk[5] = {0, 0, 0, 0, 0};
        uint64_t lt_last = __builtin_readcyclecounter();
#define LIK_TICK(i)                                       \
    {                                                     \
        const uint64_t now_ = __builtin_readcyclecounter(); \
        lk[i] += now_ - lt_last;                          \
        lt_last = now_;                                   \
    }
#else
#define LIK_TICK(i)
#endif
        MhDev Mt = M;                                        // (mh_propose_ranges reads the sub tree sizes and the braces' pointers)
        Mt.size = tb_size;
        // The slots the written nodes of the transform A feed -- a written node's own slot, and its daughters' when its height is written
        // (all: every slot) -- each once: the first lane to raise the slot's mark to this list's tag lists it.  Slots only: the new
        // distances follow once the proposal is there.  Returns the count, -1 when the list cannot hold them.
        // (the transform's integer fields as scalars by value: a struct passed around here ends up in scratch memory, its fields picked by an
        // indexed load)
        struct SegRanges {
            int kind, hlo, hhi, hlo2, hhi2, rlo, rhi, pt1, pt2, rp1, rp2, rp3, brace_lo, brace_hi;
        };
        auto build_list = [&](int a_kind, int a_hlo, int a_hhi, int a_hlo2, int a_hhi2, int a_rlo, int a_rhi, int a_pt1, int a_pt2, int a_rp1, int a_rp2, int a_rp3, int a_brace_lo, int a_brace_hi, bool all, int tagf) __attribute__((always_inline)) -> int {
            const SegRanges A{a_kind, a_hlo, a_hhi, a_hlo2, a_hhi2, a_rlo, a_rhi, a_pt1, a_pt2, a_rp1, a_rp2, a_rp3, a_brace_lo, a_brace_hi};
            int cnt = 0;
            auto emit_slot = [&](bool active, int slot_) {   // (every lane calls it: the ballots are the wave's)
                if (__builtin_amdgcn_ballot_w64(active) == 0) return;
                const int slot = active ? slot_ : -1;
                bool mine = false;
                // (a maximum, not an exchange: a slot that comes again in this list -- a written node's daughter that is written itself --
                // must keep the list position its first visit left in the mark; the tags count upwards within a launch)
                if (slot >= 0) mine = (atomicMax(&mark[slot], tagf << 8) >> 8) != tagf;
                const uint64_t mk = __builtin_amdgcn_ballot_w64(mine);
                if (mine) {
                    const int pos = cnt + (int)__builtin_popcountll(mk & lt_mask);
                    if (pos < kSsegList) {
                        l_j[pos] = slot;
                        mark[slot] = (tagf << 8) | pos;
                    }
                }
                cnt += (int)__builtin_popcountll(mk);
            };
            auto emit = [&](bool active, int node_) { emit_slot(active, active ? (int)ts_of[node_] : -1); };
            auto emit_height = [&](bool active, int w) {     // a node whose height is written: its branch and its daughters'
                emit(active, w);
                const int nc = active ? tb_nch[w] : 0;
                emit(nc > 0, active ? tb_first[w] : 0);
                emit(nc > 1, active ? tb_second[w] : 0);
            };
            if (all) {
                for (int j0 = 0; j0 < n; j0 += 64) emit_slot(j0 + lane < n, j0 + lane);
            } else {
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
            }
            if (cnt > kSsegList) cnt = -1;                   // (cannot happen for a proposal mh_capi.cpp put into a segment: the chain wave says so)
            __builtin_amdgcn_s_waitcnt(0xc07f);              // the list is in LDS before any lane reads it
            __builtin_amdgcn_wave_barrier();
            return cnt;
        };
        // the new distance of list entry m from the proposed state (LDS): likelihoodFunctionWrapper, distances = (tH * rMu) * sumFirstTwo
        // (times * rates) (app/Probability.hs:195-207) -- the arithmetic of load_tree (mvn_device.hpp) and of k_mh_step_wg's X1
        auto new_distance = [&](int slot, double s1) __attribute__((always_inline)) -> double {
            const int a = ts_node[slot], pa = ts_parent[slot];
            double x = (Hp[pa] - Hp[a]) * Rp[a];
            if (slot == 0) x = x + (Hp[0] - Hp[rr]) * Rp[rr];
            return x * s1;
        };
        constexpr int W = kSparseEllW;
        int p_cur = sched[0];
        const int vz = mh_vzero();                           // (mh_device.hpp: what travels ahead is loaded by vector loads)
        int p_next = sched[(n_steps > 1 ? 1 : 0) + vz];      // (the schedule's entries two steps ahead: a row's loads need its index)
        int kind_cur = M.kind[p_cur], node_cur = M.node[p_cur];
        for (int64_t gs = 0; gs < n_steps; ++gs) {
            const int tag = (int)gs + 1;
            const int p_next2 = sched[((gs + 2 < n_steps) ? gs + 2 : gs) + vz];
            const int kind_next = M.kind[p_next], node_next = M.node[p_next];      // (the next step's row travels while this step computes)
            // ---- AHEAD of the request, while the chain wave draws the proposal: which nodes the proposal writes follows from its table row
            // and the topology alone (mh_propose_ranges), hence the list of moved slots -- and with the list the current distances of the
            // slots, their rows of Ps and the current distances of the rows' columns: both round trips to memory are over, or under way,
            // when the request comes.  The guess is compared with the transform the proposal posts; the rare mismatch (a proposal that
            // bails out on an invalid state, u = 1 exactly) lists again.  Lists of more than 64 slots take the passes of the slow path.
            PropApply G;                                     // (its integer fields only: which nodes are written)
            mh_propose_ranges(Mt, kind_cur, node_cur, G);
            int c_kind = G.kind, c_hlo = G.hlo, c_hhi = G.hhi, c_hlo2 = G.hlo2, c_hhi2 = G.hhi2, c_rlo = G.rlo, c_rhi = G.rhi, c_pt1 = G.pt1, c_pt2 = G.pt2, c_rp1 = G.rp1, c_rp2 = G.rp2, c_rp3 = G.rp3, c_brace_lo = G.brace_lo, c_brace_hi = G.brace_hi;
            bool c_moves = seg_moves_likelihood(kind_cur, node_cur);
            bool c_all = list_all != 0 && mh_moves_scale(kind_cur, node_cur);
            int cnt = 0, tag_used = 0;
            bool pre = true;
            int jm = 0;                                      // this lane's list entry (one pass: entry m = lane)
            int k[W];
            double v[W], mu[W], x0[W], xold = 0.0;
            int more = 0;
            double s1 = s_cur;
            // (pass 0: the guess, ahead of the request; pass 1, rare: the list again from the posted transform, with a tag of its own --
            // the marks of the guess are stale)
            for (int pass = 0; pass < 2; ++pass) {
                tag_used = 2 * tag + pass;
                cnt = c_moves ? build_list(c_kind, c_hlo, c_hhi, c_hlo2, c_hhi2, c_rlo, c_rhi, c_pt1, c_pt2, c_rp1, c_rp2, c_rp3, c_brace_lo, c_brace_hi, c_all, tag_used) : 0;
                pre = cnt >= 0 && cnt <= 64;
                if (pre && cnt > 0) {
                    // entry m = lane of a list of at most 64 slots: its slot's current distance and row of Ps, then the current distances
                    // of the row's columns
                    const bool act = lane < cnt;
                    jm = act ? l_j[lane] : 0;
                    xold = X0[jm];
                    more = Sp.ell_more[jm];
                    const int4* pc = reinterpret_cast<const int4*>(Sp.ell_col + (size_t)jm * W);
                    const d2* pv = reinterpret_cast<const d2*>(Sp.ell_val + (size_t)jm * W);
                    const d2* pm = reinterpret_cast<const d2*>(Sp.ell_mu + (size_t)jm * W);
#pragma unroll
                    for (int u = 0; u < W / 4; ++u) {
                        const int4 c4 = pc[u];
                        k[4 * u] = c4.x;
                        k[4 * u + 1] = c4.y;
                        k[4 * u + 2] = c4.z;
                        k[4 * u + 3] = c4.w;
                    }
#pragma unroll
                    for (int u = 0; u < W / 2; ++u) {
                        const d2 a2 = pv[u], c2 = pm[u];
                        v[2 * u] = a2.x;
                        v[2 * u + 1] = a2.y;
                        mu[2 * u] = c2.x;
                        mu[2 * u + 1] = c2.y;
                    }
#pragma unroll
                    for (int u = 0; u < W; ++u) x0[u] = X0[k[u]];      // (padding entries: the row's own index, weight 0)
                }
                if (pass == 1) break;
                LIK_TICK(1)
                (void)seg_poll(w_req, tag, 0);
                LIK_TICK(0)
                const bool moves = *w_moves != 0;
                s1 = moves ? (double)*w_s1 : s_cur;
                const bool all = list_all != 0 && moves && s1 != s_cur;
                // (the posted transform's integer fields)
                const int n_kind = A_lds->kind, n_hlo = A_lds->hlo, n_hhi = A_lds->hhi, n_hlo2 = A_lds->hlo2, n_hhi2 = A_lds->hhi2, n_rlo = A_lds->rlo, n_rhi = A_lds->rhi, n_pt1 = A_lds->pt1, n_pt2 = A_lds->pt2, n_rp1 = A_lds->rp1, n_rp2 = A_lds->rp2, n_rp3 = A_lds->rp3, n_brace_lo = A_lds->brace_lo, n_brace_hi = A_lds->brace_hi;
                const bool same_ranges = n_kind == c_kind && n_hlo == c_hlo && n_hhi == c_hhi && n_hlo2 == c_hlo2 && n_hhi2 == c_hhi2 && n_rlo == c_rlo && n_rhi == c_rhi && n_pt1 == c_pt1 && n_pt2 == c_pt2 && n_rp1 == c_rp1 && n_rp2 == c_rp2 && n_rp3 == c_rp3 && n_brace_lo == c_brace_lo && n_brace_hi == c_brace_hi;
                if (pre && moves == c_moves && (!moves || (same_ranges && all == c_all))) break;
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
                c_all = all;
            }
            double contrib = 0.0;
            bool have0 = false;
            double d0 = 0.0;
            if (pre) {
                // ---- one pass: entry m = lane
                const bool act = lane < cnt;
                const double xn = act ? new_distance(jm, s1) : 0.0;
                const double dj = act ? xn - xold : 0.0;
                if (act) {
                    l_dnew[lane] = xn;
                    l_delta[lane] = dj;
                    if (jm == 0) {
                        have0 = true;
                        d0 = xn;
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);          // every lane's delta is in LDS before any lane looks one up
                __builtin_amdgcn_wave_barrier();
                if (cnt > 0) {
                    // q' - q = sum_{j in J} delta_j sum_k Ps[j][k] (2 dx_k + delta_k); the look-ups without a branch: all marks, then all deltas
                    int mk[W];
                    double dk[W];
#pragma unroll
                    for (int u = 0; u < W; ++u) mk[u] = mark[k[u]];
#pragma unroll
                    for (int u = 0; u < W; ++u) dk[u] = l_delta[mk[u] & 0xFF];
                    double acc = 0.0;
#pragma unroll
                    for (int u = 0; u < W; ++u) acc = fma(v[u], 2.0 * (x0[u] - mu[u]) + (((mk[u] >> 8) == tag_used) ? dk[u] : 0.0), acc);
                    if (more > 0 && act) {                   // a row longer than the record: the rest from the CSR arrays
                        const int p1 = Sp.s_rowptr[jm + 1];
                        for (int e = p1 - more; e < p1; ++e) {
                            const int kk = Sp.s_col[e];
                            const int m2 = mark[kk];
                            const double d2k = ((m2 >> 8) == tag_used) ? l_delta[m2 & 0xFF] : 0.0;
                            acc = fma(Sp.s_val[e], 2.0 * (X0[kk] - Sp.mu[kk]) + d2k, acc);
                        }
                    }
                    contrib = fma(dj, acc, contrib);
                }
            } else if (cnt > 64) {
                // ---- more than 64 slots: every delta first, then the rows in passes of 64 entries
                for (int m = lane; m < cnt; m += 64) {
                    const int j = l_j[m];
                    const double xn = new_distance(j, s1);
                    l_dnew[m] = xn;
                    l_delta[m] = xn - X0[j];
                    if (j == 0) {
                        have0 = true;
                        d0 = xn;
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                for (int m0 = 0; m0 < cnt; m0 += 64) {
                    const int m = m0 + lane;
                    const bool act = m < cnt;
                    const int j = act ? l_j[m] : 0;
                    const double dj = act ? l_delta[m] : 0.0;
                    const int p0 = Sp.s_rowptr[j], p1 = Sp.s_rowptr[j + 1];
                    double acc = 0.0;
                    for (int e = p0; e < p1; ++e) {
                        const int kk = Sp.s_col[e];
                        const int m2 = mark[kk];
                        const double d2k = ((m2 >> 8) == tag_used) ? l_delta[m2 & 0xFF] : 0.0;
                        acc = fma(Sp.s_val[e], 2.0 * (X0[kk] - Sp.mu[kk]) + d2k, acc);
                    }
                    contrib = fma(dj, acc, contrib);
                }
            }
            {
                const uint64_t m0 = __builtin_amdgcn_ballot_w64(have0);
                if (lane == 0) *w_have0 = (m0 != 0) ? 1 : 0;
                if (m0 != 0) {
                    const double lj1 = log(1.0 / mh_readlane64(d0, (int)__builtin_ctzll(m0)));     // jacobianRootBranch, :393-410
                    if (lane == 0) *w_lj = lj1;
                }
            }
            LIK_TICK(2)
            const double qp = (cnt > 0) ? q + wave_sum(contrib) : q;
            if (lane == 0) {
                *w_q = qp;
                *w_cnt = cnt;
            }
            seg_post(w_resp, tag);                           // (every lane stores the same word: the fence is the wave's)
            LIK_TICK(3)
            const int d = seg_poll(w_dec, tag, 1);
            if ((d & 1) && cnt > 0) {
                q = qp;
                s_cur = s1;
                if (valid)
                    for (int m = lane; m < cnt; m += 64) X0[l_j[m]] = l_dnew[m];
            } else if (d & 1) {
                s_cur = s1;
            }
            p_cur = p_next;
            p_next = p_next2;
            kind_cur = __builtin_amdgcn_readfirstlane(kind_next);
            node_cur = __builtin_amdgcn_readfirstlane(node_next);
            LIK_TICK(4)
        }
#ifdef MCD_SEG_STAMP
        if (trace_alpha && lane == 0 && valid)
            for (int i = 0; i < 5; ++i) trace_alpha[(int64_t)(8 + i) * B + b] = (double)lk[i];
#endif
        if (valid && lane == 0) I.zcur[b] = q;
        seg_tail_distances(M, Q, words, Hp, Rp, ts_node, ts_parent, rr, n, n_steps, b, valid, lane);
        return;
    }

    // ================================================================ chain waves (mh_segment_device.hpp: shared with the dense kernel)
    // (two instances with the prior waves: the plain loop for small trees when no prior wave draws -- MCD_MH_PRIOR_DRAWS = 0)
    if (HELP && !Q.prior_draws && nn < Q.ahead_from)
        seg_chain_wave<HELP, HELP>(M, P, Pst, L, Q, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, b, valid, lane);
    else
        seg_chain_wave<HELP, false>(M, P, Pst, L, Q, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, b, valid, lane);
}

// any tree of 3 .. 2048 nodes whose tables and one chain fit a CU's LDS, the symmetric part of the matrix on the device
bool mh_segment_sparse_available(const MhDev& M, const SparseDev& Sp)
{
    if (Sp.s_rowptr == nullptr || M.n_nodes < 3 || M.n_nodes > 2048 || Sp.n != M.n_nodes - 2 || M.batch > 65536) return false;
    int cpw = 0, sz = 0;
    sseg_geometry(M.n_nodes, (Sp.n + 63) / 64 * 64, cpw, sz);
    return cpw > 0;
}
int mh_segment_sparse_list() { return kSsegList; }

template <int CPW, bool HELP>
static hipError_t launch_sseg(const MhDev& M, const SparseDev& Sp, const TreeDev& T, const PriorDev& P, const MhInc& I, const int32_t* sched, int64_t n_steps,
                              int32_t S, int accumulate, uint64_t step0, uint64_t seed, double* trace_alpha, int8_t* trace_accept, int64_t gs_base,
                              int summands_kept, const MhSegPending& Q, int size_in_lds, int list_all, hipStream_t st)
{
    const int np = (Sp.n + 63) / 64 * 64;
    const size_t dynb = sseg_lds_bytes(M.n_nodes, np, CPW, size_in_lds) + sseg_node_tables_bytes(M.n_nodes, np, CPW, size_in_lds, P.n_cal, P.n_con);
    static std::atomic<unsigned long long> allowed{0};       // more than 64 KiB of LDS has to be allowed once per device
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!((allowed.load(std::memory_order_acquire) >> dev) & 1ull)) {
        if (hipError_t e = hipFuncSetAttribute((const void*)k_mh_segment_sparse<CPW, HELP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSsegLdsMax)) return e;
        allowed.fetch_or(1ull << dev, std::memory_order_release);
    }
    note_dynamic_lds(dynb);
    hipLaunchKernelGGL((k_mh_segment_sparse<CPW, HELP>), dim3((unsigned)((M.batch + CPW - 1) / CPW)), dim3(64 * CPW * (HELP ? 4 : 2)), dynb, st, M, Sp, T, P, I, sched, n_steps, S, accumulate,
                       step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, size_in_lds, list_all);
    return hipGetLastError();
}

// steps [0, n_steps) of `sched` (device memory), none of which moves more than kSsegList distances; I: X0 = the current distances
// [batch][n], zcur / zprop = the quadratic forms q [batch] of the current states / of the pending dense proposal (NPz = 1); list_all:
// the tree's distance slots all fit the list, so a proposal that moves tH or rMu (every distance) may be part of a segment
hipError_t launch_mh_segment_sparse(const MhDev& M, const SparseDev& Sp, const TreeDev& T, const PriorDev& P, const MhInc& I, const int32_t* sched,
                                    int64_t n_steps, int32_t S, int accumulate, uint64_t step0, uint64_t seed, double* trace_alpha, int8_t* trace_accept,
                                    int64_t gs_base, int summands_kept, const MhSegPending* pending, int list_all, hipStream_t st)
{
    MhSegPending Q{};
    Q.p_acc = -1;
    Q.p_tail = -1;
    if (pending) Q = *pending;
    Q.ahead_from = opt_or(OPT_MH_AHEAD_FROM, kSegAheadFrom);
    Q.prior_draws = (!opt_is(OPT_MH_PRIOR_DRAWS, 0) && !opt_is(OPT_MH_PRIOR_WAVES, 0)) ? 1 : 0;
    if (n_steps <= 0) return Q.p_acc >= 0 ? hipErrorInvalidValue : hipSuccess;
    if (Q.p_acc >= 0 && (Q.X1 == nullptr || I.zprop == nullptr || !summands_kept)) return hipErrorInvalidValue;
    if (Q.p_tail >= M.n_prop || (Q.p_tail >= 0 && (Q.X1_tail == nullptr || M.psum == nullptr || M.psel == nullptr))) return hipErrorInvalidValue;
    if (n_steps > (1 << 22)) return hipErrorInvalidValue;    // (a slot's mark holds the step in 23 bits)
    if (!mh_segment_sparse_available(M, Sp) || I.X0 == nullptr || I.zcur == nullptr || I.NPz != 1) return hipErrorInvalidValue;
    if (list_all && Sp.n > kSsegList) return hipErrorInvalidValue;
    int cpw = 0, sz = 0;
    sseg_geometry(M.n_nodes, (Sp.n + 63) / 64 * 64, cpw, sz);
    const bool help = !opt_is(OPT_MH_PRIOR_WAVES, 0);        // (mcd_set_option "MCD_MH_PRIOR_WAVES" = 0: the chain wave evaluates the whole ln prior; tests, timing)
    if (cpw == 2 && help) return launch_sseg<2, true>(M, Sp, T, P, I, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, sz, list_all, st);
    if (cpw == 2) return launch_sseg<2, false>(M, Sp, T, P, I, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, sz, list_all, st);
    if (help) return launch_sseg<1, true>(M, Sp, T, P, I, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, sz, list_all, st);
    return launch_sseg<1, false>(M, Sp, T, P, I, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, sz, list_all, st);
}

}  // namespace mcd
