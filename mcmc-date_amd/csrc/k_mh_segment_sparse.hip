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

#include <atomic>

namespace mcd {

constexpr int kSsegList = 254;           // moved distances of one proposal at most: a slot's mark holds its list position in 8 bits
constexpr int kSsegListAlloc = 256;
constexpr int kSsegRow = 16;             // entries of a row of Ps in flight per lane

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
    return 6 * (size_t)n_nodes + 2 * (size_t)kSsegListAlloc + (size_t)kSsegListAlloc / 2 + (size_t)np / 2 + 8 + (size_t)kSegApplyDoubles;
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

template <int CPW>
__global__ __launch_bounds__(128 * CPW, 1) void k_mh_segment_sparse(MhDev M, SparseDev Sp, TreeDev T, PriorDev P, MhInc I, const int32_t* __restrict__ sched,
                                                                   int64_t n_steps, int32_t S, int accumulate, uint64_t step0, uint64_t seed,
                                                                   double* __restrict__ trace_alpha, int8_t* __restrict__ trace_accept, int64_t gs_base,
                                                                   int summands_kept, MhSegPending Q, int size_in_lds, int list_all)
{
    extern __shared__ double dyn[];
    constexpr int NT = 128 * CPW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cs = wave % CPW;                               // which of the workgroup's chains
    const bool chain_role = wave < CPW;
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

    // ================================================================ likelihood waves
    if (!chain_role) {
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
        for (int64_t gs = 0; gs < n_steps; ++gs) {
            const int tag = (int)gs + 1;
            (void)seg_poll(w_req, tag, 0);
            // ---- the distances the written nodes feed: a written node's own slot, and its daughters' when its height is written.  Each
            // slot once (the first lane to exchange the slot's mark for this step's lists it), with its new distance.
            int cnt = 0;
            double s1 = s_cur;
            if (*w_moves) {
                const PropApply A = *A_lds;
                s1 = *w_s1;
                double d0 = 0.0;                             // the new distance of slot 0, in the lane that listed it
                bool have0 = false;
                auto emit_slot = [&](bool active, int slot_) {    // (every lane calls it: the ballots are the wave's)
                    if (__builtin_amdgcn_ballot_w64(active) == 0) return;
                    const int slot = active ? slot_ : -1;
                    bool mine = false;
                    // (a maximum, not an exchange: a slot that comes again in this step -- a written node's daughter that is written itself --
                    // must keep the list position its first visit left in the mark; steps count upwards within a launch)
                    if (slot >= 0) mine = (atomicMax(&mark[slot], tag << 8) >> 8) != tag;
                    const uint64_t mk = __builtin_amdgcn_ballot_w64(mine);
                    if (mine) {
                        const int pos = cnt + (int)__builtin_popcountll(mk & lt_mask);
                        // likelihoodFunctionWrapper: distances = (tH * rMu) * sumFirstTwo (times * rates)   (app/Probability.hs:195-207), the
                        // arithmetic of load_tree (mvn_device.hpp) and of k_mh_step_wg's X1
                        const int a = ts_node[slot], pa = ts_parent[slot];
                        double x = (Hp[pa] - Hp[a]) * Rp[a];
                        if (slot == 0) x = x + (Hp[0] - Hp[rr]) * Rp[rr];
                        x = x * s1;
                        if (pos < kSsegList) {
                            l_j[pos] = slot;
                            l_dnew[pos] = x;
                            mark[slot] = (tag << 8) | pos;
                        }
                        if (slot == 0) {
                            d0 = x;
                            have0 = true;
                        }
                    }
                    cnt += (int)__builtin_popcountll(mk);
                };
                auto emit = [&](bool active, int node_) { emit_slot(active, active ? (int)ts_of[node_] : -1); };
                auto emit_height = [&](bool active, int w) {    // a node whose height is written: its branch and its daughters'
                    emit(active, w);
                    const int nc = active ? tb_nch[w] : 0;
                    emit(nc > 0, active ? tb_first[w] : 0);
                    emit(nc > 1, active ? tb_second[w] : 0);
                };
                if (list_all && s1 != s_cur) {
                    // tH or rMu moved: every distance did (a tree whose slots all fit the list; mh_capi.cpp puts such a proposal into a
                    // segment only then)
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
                        emit_height(lane == 0, x);           // (SLIDE_BRACE_CONTRA also writes the rates of x and its daughters: the same slots)
                    }
                }
                const uint64_t m0 = __builtin_amdgcn_ballot_w64(have0);
                if (lane == 0) *w_have0 = (m0 != 0) ? 1 : 0;
                if (m0 != 0) {
                    const double lj1 = log(1.0 / mh_readlane64(d0, (int)__builtin_ctzll(m0)));     // jacobianRootBranch, :393-410
                    if (lane == 0) *w_lj = lj1;
                }
                if (cnt > kSsegList) cnt = -1;               // (cannot happen for a proposal mh_capi.cpp put into a segment: the chain wave says so)
                __builtin_amdgcn_s_waitcnt(0xc07f);          // the list is in LDS before any lane reads it
                __builtin_amdgcn_wave_barrier();
                // the deltas against the current distances: one batch of loads
                for (int m = lane; m < cnt; m += 64) l_delta[m] = l_dnew[m] - X0[l_j[m]];
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
            // ---- q' - q = sum_{j in J} delta_j sum_k Ps[j][k] (2 dx_k + delta_k): lanes = listed slots, each walks its row of Ps
            double contrib = 0.0;
            for (int m = lane; m < cnt; m += 64) {
                const int j = l_j[m];
                const double dj = l_delta[m];
                const int p0 = Sp.s_rowptr[j], p1 = Sp.s_rowptr[j + 1];
                double acc = 0.0;
                for (int e0 = p0; e0 < p1; e0 += kSsegRow) {
                    int k[kSsegRow];
                    double v[kSsegRow], x0[kSsegRow], mu[kSsegRow];
#pragma unroll
                    for (int u = 0; u < kSsegRow; ++u) {
                        const int e = (e0 + u < p1) ? e0 + u : p1 - 1;       // (past the row's end: its last entry again, weight 0)
                        k[u] = Sp.s_col[e];
                        v[u] = (e0 + u < p1) ? Sp.s_val[e] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < kSsegRow; ++u) {
                        x0[u] = X0[k[u]];
                        mu[u] = Sp.mu[k[u]];
                    }
#pragma unroll
                    for (int u = 0; u < kSsegRow; ++u) {
                        const int mk = mark[k[u]];
                        const double dk = ((mk >> 8) == tag) ? l_delta[mk & 0xFF] : 0.0;
                        acc = fma(v[u], 2.0 * (x0[u] - mu[u]) + dk, acc);
                    }
                }
                contrib = fma(dj, acc, contrib);
            }
            const double qp = (cnt > 0) ? q + wave_sum(contrib) : q;
            if (lane == 0) {
                *w_q = qp;
                *w_cnt = cnt;
            }
            seg_post(w_resp, tag);                           // (every lane stores the same word: the fence is the wave's)
            const int d = seg_poll(w_dec, tag, 1);
            if ((d & 1) && cnt > 0) {
                q = qp;
                s_cur = s1;
                if (valid)
                    for (int m = lane; m < cnt; m += 64) X0[l_j[m]] = l_dnew[m];
            } else if (d & 1) {
                s_cur = s1;
            }
        }
        if (valid && lane == 0) I.zcur[b] = q;
        return;
    }

    // ================================================================ chain waves (mh_segment_device.hpp: shared with the dense kernel)
    SegChainCtx L;
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
    L.c = Sp.c;
    L.logdet = Sp.logdet;
    seg_chain_wave(M, P, Pst, L, Q, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, b, valid, lane);
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

template <int CPW>
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
        if (hipError_t e = hipFuncSetAttribute((const void*)k_mh_segment_sparse<CPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSsegLdsMax)) return e;
        allowed.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL(k_mh_segment_sparse<CPW>, dim3((unsigned)((M.batch + CPW - 1) / CPW)), dim3(128 * CPW), dynb, st, M, Sp, T, P, I, sched, n_steps, S, accumulate,
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
    if (pending) Q = *pending;
    if (n_steps <= 0) return Q.p_acc >= 0 ? hipErrorInvalidValue : hipSuccess;
    if (Q.p_acc >= 0 && (Q.X1 == nullptr || I.zprop == nullptr || !summands_kept)) return hipErrorInvalidValue;
    if (n_steps > (1 << 22)) return hipErrorInvalidValue;    // (a slot's mark holds the step in 23 bits)
    if (!mh_segment_sparse_available(M, Sp) || I.X0 == nullptr || I.zcur == nullptr || I.NPz != 1) return hipErrorInvalidValue;
    if (list_all && Sp.n > kSsegList) return hipErrorInvalidValue;
    int cpw = 0, sz = 0;
    sseg_geometry(M.n_nodes, (Sp.n + 63) / 64 * 64, cpw, sz);
    if (cpw == 2) return launch_sseg<2>(M, Sp, T, P, I, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, sz, list_all, st);
    return launch_sseg<1>(M, Sp, T, P, I, sched, n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept, gs_base, summands_kept, Q, sz, list_all, st);
}

}  // namespace mcd
