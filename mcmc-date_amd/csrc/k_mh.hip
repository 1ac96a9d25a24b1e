// k_mh.hip -- lock-step Metropolis-Hastings-Green steps for a batch of independent chains (gfx950).
// SURVEY.md 8(f) row f2.  All chains execute the same proposal of the cycle at the same time, each with its own
// counter-based random numbers, tuning parameter and accept/reject decision; the posterior of the proposed states
// is evaluated by the prior code inside `k_mh_step` and by the batched likelihood kernel between two `k_mh_step` launches.
//
// Proposals restated here (paths relative to the reference):
//   slide node / scale sub tree / pulley, ultrametric   lib/Mcmc/Tree/Proposal/Ultrametric.hs:50-59, 126-149, 221-286
//   truncatedNormalSample                               lib/Mcmc/Tree/Proposal/Internal.hs:107-138
//   truncated normal density / quantile                 lib/Statistics/Distribution/TruncatedNormal.hs:55-130
//   scale branch / scale (sub) tree of rates            lib/Mcmc/Tree/Proposal/Unconstrained.hs:40-130
//   scaleNormAndTreeContrarily                          .../Unconstrained.hs:221-256
//   scaleVarianceAndTree (and the autocorrelated form)  .../Unconstrained.hs:286-316, 354-386
//   scaleUnbiased / scaleContrarily, MHG acceptance, auto tuning: package `mcmc` [external, dschrempf/mcmc 542c43f6]
//   liftProposalWith jacobianRootBranch                 app/Definitions.hs:148 ff., app/Probability.hs:393-410
//
// Mapping: one wave per chain, lanes = nodes (strided for trees with more than 64 nodes).  Pre-order numbering makes
// every sub tree a contiguous index range [v, v + size[v]), so "scale the sub tree" is a range test per lane.  The
// scalar part of a proposal (random draw, bounds, ratio) is computed redundantly by all lanes of the wave.
// Random numbers: Philox4x32-10, counter = (draw, chain, step_lo, step_hi), key = seed; the proposal's draws use
// draw = 0, 1, ..., the acceptance uniform is draw 0xFFFFFFFF, the boost of a gamma with shape < 1 is 0xFFFFFFFE.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "mh_inc_device.hpp"
#include "options.h"

#include <atomic>
#include <type_traits>
#include "prior_device.hpp"

namespace mcd {

// diagnostic build (make stamp_mhstep): s_memtime ticks (shader cycles) per phase of wave 0 of workgroup 0, summed per proposal
// kind; read back with mcd_mhstep_debug_stamps (tools/microbench/mhstep_stamps.py)
#ifdef MCD_MHSTEP_STAMP
__device__ unsigned long long g_mhs_acc[32 * 10];
__device__ unsigned long long g_mhs_cnt[32];
#define MHS_T(i) do { mhs[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MHS_T(i) do { } while (0)
#endif

// One launch between two likelihood launches: ACCEPT the pending step (proposal p_acc; its proposed state, ln prior, ln
// likelihood and ln jacobianRootBranch are in H1/R1/sc1/post1) and PROPOSE the next one (proposal p_prop, its table row
// passed by value) together with the ln prior of its proposed state.  Either half is skipped with a negative proposal
// id (first / last launch of a run).  One wave per chain; the current
// and the proposed heights and rates are staged in LDS (one region per wave), so no lane ever reads through global memory
// what another lane of its wave has just written.
//   accept:  post = (ln prior, ln likelihood, ln jacobianRootBranch), [3][batch]; counters; optional trace; running sums
//            of the absolute node ages tH * h_v when the step closes an iteration (`accumulate_now`)
//   propose: writes sc1, H1, R1, lnqj (ln q-ratio * Jacobian without the root-branch factor), post1[0] = ln prior
__global__ __launch_bounds__(256) void k_mh_step(MhDev M, PriorDev P, int p_acc, int jac_root_acc, int p_prop, PropRow row_prop,
                                                 int draw_slot, uint64_t step_acc, uint64_t seed, int accumulate_now,
                                                 double* __restrict__ trace_alpha, int8_t* __restrict__ trace_accept, int prior_inline)
{
    extern __shared__ double sh[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (b >= M.batch) return;                             // wave-uniform; no workgroup barriers in this kernel
    const int n = M.n_nodes;
    const int64_t B = M.batch;
    double* Hc = sh + (size_t)wave * 4 * n;
    double* Rc = Hc + n;
    double* Hs = Rc + n;
    double* Rs = Hs + n;
    bool ok = false;
#ifdef MCD_MHSTEP_STAMP
    unsigned long long mhs[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    MHS_T(0);
    if (p_acc >= 0) {
        const int p = p_acc;
        const double lp = M.post[b], ll = M.post[B + b], lj = M.post[2 * B + b];
        const double lp1 = M.post1[b], ll1 = M.post1[B + b], lj1 = M.post1[2 * B + b];
        double la = M.beta[b] * ((lp1 + ll1) - (lp + ll)) + M.lnqj[b];     // heated chains of MC3: posterior^beta; beta = 1 is exact
        if (jac_root_acc) la += (double)jac_root_acc * (lj1 - lj);   // +1: jf(y) / jf(x); -1 (experiments): the reciprocal
        double ua, ub;
        philox_block(mh_rng(seed, M.chain0 + b, step_acc), 0xFFFFFFFFu, ua, ub);
        ok = (la >= 0) || (ua < exp(la));
        if (lane == 0) {
            const int64_t i = b * M.n_prop + p;
            M.tried[i] += 1;
            if (ok) M.acc[i] += 1;
            if (trace_alpha) trace_alpha[b] = la;
            if (trace_accept) trace_accept[b] = ok ? 1 : 0;
        }
    }
    MHS_T(1);
    // the current state after the decision, into LDS (and back to global memory when it changed)
    const double* Hsrc = (ok ? M.H1 : M.H) + b * M.ld;
    const double* Rsrc = (ok ? M.R1 : M.R) + b * M.ld;
    for (int w = lane; w < n; w += 64) {
        const double h = Hsrc[w], r = Rsrc[w];
        Hc[w] = h;
        Rc[w] = r;
        if (ok) {
            M.H[b * M.ld + w] = h;
            M.R[b * M.ld + w] = r;
        }
    }
    double sc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) sc[i] = (ok ? M.sc1 : M.sc)[i * B + b];
    if (ok) {
        if (lane < 5) {
            double mine = sc[0];
#pragma unroll
            for (int i = 1; i < 5; ++i)
                if (lane == i) mine = sc[i];
            M.sc[lane * B + b] = mine;
        }
        if (lane < 3) {
            M.post[lane * B + b] = M.post1[lane * B + b];
            M.pcomp[b * 3 + lane] = M.pcomp1[b * 3 + lane];
        }
    }
    if (accumulate_now) {
        for (int w = lane; w < n; w += 64) {
            const double a = sc[2] * Hc[w];
            M.age_sum[b * n + w] += a;
            M.age_sq[b * n + w] += a * a;
        }
    }
    if (p_prop < 0) return;
    __builtin_amdgcn_wave_barrier();
    MHS_T(2);
    const double t = M.tune[b * M.n_prop + p_prop];
    // the state-independent draws of this step were computed by k_mh_draws, one thread per (step, chain)
    const double* dw = M.draws + ((size_t)draw_slot * B + b) * 5;
    const StepDraws dr{dw[0], dw[1], dw[2], dw[3], dw[4]};
    (void)seed;
    double sc0[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) sc0[i] = sc[i];
    const double lnqj = mh_propose_wave(M, row_prop, t, dr, lane, sc, Hc, Rc, Hs, Rs);
    __builtin_amdgcn_wave_barrier();
    MHS_T(3);
    // re-evaluate only the blocks of the ln prior whose inputs the proposal moved (as k_mh_chain.hip does)
    bool dH = false, dR = false;
    for (int w0 = 0; w0 < n; w0 += 64) {
        const int w = w0 + lane;
        const bool in = w < n;
        dH = dH || (__builtin_amdgcn_ballot_w64(in && Hs[in ? w : 0] != Hc[in ? w : 0]) != 0);
        dR = dR || (__builtin_amdgcn_ballot_w64(in && Rs[in ? w : 0] != Rc[in ? w : 0]) != 0);
    }
    MHS_T(4);
    const bool f0 = dH || sc[2] != sc0[2];
    const bool f1 = dH || sc[0] != sc0[0] || sc[1] != sc0[1];
    const bool f2 = dR || sc[3] != sc0[3] || sc[4] != sc0[4] || (dH && P.clock_model >= 2);
    if (!prior_inline) {
        // the ln prior of the proposed state is evaluated beside its ln likelihood (mh_prior_role.hpp): leave what that needs
        for (int w = lane; w < n; w += 64) {
            M.H1[b * M.ld + w] = Hs[w];
            M.R1[b * M.ld + w] = Rs[w];
        }
        if (lane < 5) {
            double mine = sc[0];
#pragma unroll
            for (int i = 1; i < 5; ++i)
                if (lane == i) mine = sc[i];
            M.sc1[lane * B + b] = mine;
        }
        if (lane == 0) {
            M.lnqj[b] = lnqj;
            M.pflags[b] = (f0 ? 1 : 0) | (f1 ? 2 : 0) | (f2 ? 4 : 0);
        }
        return;
    }
    const double c0p = (dH || sc[2] != sc0[2]) ? prior_nodes_wave(P, lane, sc[2], Hs) : M.pcomp[b * 3 + 0];
    MHS_T(5);
    const double c1p = (dH || sc[0] != sc0[0] || sc[1] != sc0[1]) ? prior_bd_wave(P, lane, sc[0], sc[1], Hs) : M.pcomp[b * 3 + 1];
    MHS_T(6);
    const double c2p = (dR || sc[3] != sc0[3] || sc[4] != sc0[4] || (dH && P.clock_model >= 2)) ? prior_clock_wave(P, lane, sc[3], sc[4], Hs, Rs)
                                                                                             : M.pcomp[b * 3 + 2];
    const double lp1 = c0p + c1p + c2p;
    MHS_T(7);
    if (lane == 0) {
        M.pcomp1[b * 3 + 0] = c0p;
        M.pcomp1[b * 3 + 1] = c1p;
        M.pcomp1[b * 3 + 2] = c2p;
    }
    for (int w = lane; w < n; w += 64) {
        M.H1[b * M.ld + w] = Hs[w];
        M.R1[b * M.ld + w] = Rs[w];
    }
    if (lane < 5) {
        double mine = sc[0];
#pragma unroll
        for (int i = 1; i < 5; ++i)
            if (lane == i) mine = sc[i];
        M.sc1[lane * B + b] = mine;
    }
    if (lane == 0) {
        M.lnqj[b] = lnqj;
        M.post1[b] = lp1;
    }
#ifdef MCD_MHSTEP_STAMP
    __builtin_amdgcn_s_waitcnt(0);
    MHS_T(8);
    if (b == 0 && lane == 0) {
        const int k = row_prop.kind & 31;
        for (int i = 0; i < 8; ++i) g_mhs_acc[k * 10 + i] += mhs[i + 1] - mhs[i];
        g_mhs_cnt[k] += 1;
    }
#endif
}

// The same step with a WORKGROUP of four waves per chain, for trees of more than 320 nodes: there one wave walks 6 .. 17 strides of
// 64 nodes through every loop -- state copy, comparison, stores, and above all the two blocks of the ln prior, 16 trips through
// the exponentials and logarithms each at 1025 nodes.  Wave 0 is the chain's wave: it runs the wave-level proposal code and closes
// the sums.  All four waves copy the state and write the proposal back (threads = nodes) and evaluate the per-node
// summands of the birth-death and the clock block, the 64-node iterations dealt round-robin, into LDS, and wave 0 adds them lane
// by lane in the order of the iterations and then over the wave, exactly as prior_bd_wave / prior_clock_wave do alone: the same
// bits as k_mh_step.  (At 257 nodes the two forms take the same time -- one 64-node iteration of summands costs a wave as much
// alone as three of four pipelined, profiles/r02_mhstep_phases.txt -- so the smaller trees keep the one-wave kernel.)
constexpr int MHW = 4;                                     // waves per chain (two workgroups per CU at two waves per SIMD: 256 VGPRs each, no spills)
// Global memory is a microsecond away and a launch of this kernel is a dependent chain of a dozen phases, so every phase issues ALL
// its loads before it waits for one: KM = ceil(n_nodes / 256) elements per thread in registers, statically unrolled (a loop that
// loads, waits and stores per trip is a round trip per trip: the state copy alone was five of them at 1025 nodes).  The tree's index
// arrays are staged in LDS once per launch for the same reason.
template <int KM>
__global__ __launch_bounds__(64 * MHW, 2) void k_mh_step_wg(MhDev M, PriorDev P, int p_acc, int jac_root_acc, int p_prop, PropRow row_prop,
                                                            int draw_slot, uint64_t step_acc, uint64_t seed, int accumulate_now,
                                                            double* __restrict__ trace_alpha, int8_t* __restrict__ trace_accept, int prior_inline,
                                                            TreeDev T, int n_dim, double* __restrict__ X1, int64_t ldx, MhInc I, MvnDev V,
                                                            int summands_init)
{
    __shared__ IncShared incsh;
    extern __shared__ double sh[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NT = 64 * MHW;
    constexpr int KZ = 4;                                  // rows of z per thread (incremental evaluation: at most 1024 rows)
    const int64_t b = blockIdx.x;
    const int n = M.n_nodes;
    const int64_t B = M.batch;
    const int NIT = (n - 1 + 63) >> 6;                     // iterations of v = 1 + lane + 64 it < n
    const int NS = NIT * 64;
    double* Hc = sh;                                       // current state after the decision
    double* Rc = Hc + n;
    double* Hs = Rc + n;                                   // proposed state
    double* Rs = Hs + n;
    double* tb = Rs + n;                                   // [NIT][64] summands of the birth-death block
    double* tc = tb + NS;                                  // [NIT][64] summands of the clock block
    double* bc = tc + NS;                                  // [48] wave 0 -> all: proposed scalars, flags, the per-node transform; workers -> wave 0: c0, hyper
    int* par = reinterpret_cast<int*>(bc + 48);            // [n] parent of a node
    int* snode = par + n;                                  // [n_dim] node and parent of a distance slot
    int* spar = snode + n;
    const bool dist = X1 != nullptr;
    const bool incr = dist && I.X0 != nullptr;
    const bool inc_cols = incr && I.prop_mode == 1;        // this launch evaluates its sparse proposal itself (columns of L^-1; dense likelihood, at most 1024 slots)
    const bool cached = prior_inline && M.psum != nullptr;
    const bool cache_init = summands_init != 0;           // the first launch of a run: no summands kept yet
#ifdef MCD_MHSTEP_STAMP
    unsigned long long mhs[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define MHS_W(i) do { __builtin_amdgcn_s_waitcnt(0); MHS_T(i); } while (0)      /* (diagnostic build: the loads so far have landed) */
#else
#define MHS_W(i) do { } while (0)
#endif
    MHS_T(0);
    // ---- every scalar either outcome of the decision needs, then the tree's index arrays: one batch of loads
    double po[3] = {0.0, 0.0, 0.0}, po1[3] = {0.0, 0.0, 0.0}, be = 1.0, lq = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) po[i] = M.post[i * B + b];
    if (p_acc >= 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) po1[i] = M.post1[i * B + b];
        be = M.beta[b];
        lq = M.lnqj[b];
    }
    double scA[5], scB[5], pcA[3], pcB[3];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        scA[i] = M.sc[i * B + b];
        scB[i] = (p_acc >= 0) ? M.sc1[i * B + b] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        pcA[i] = M.pcomp[b * 3 + i];
        pcB[i] = (p_acc >= 0) ? M.pcomp1[b * 3 + i] : 0.0;
    }
    int2 sel_pf = make_int2(0, 0);                          // which buffers hold the current summands; the blocks the pending proposal wrote
    if (cached && !cache_init) sel_pf = reinterpret_cast<const int2*>(M.psel)[b];
    int ix_p[KM], ix_a[KM], ix_q[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        const int w = tid + NT * k;
        ix_p[k] = (w < n) ? P.parent[w] : 0;
        ix_a[k] = (dist && w < n_dim) ? T.slot_node[w] : 0;
        ix_q[k] = (dist && w < n_dim) ? T.slot_parent[w] : 0;
    }
    MHS_W(1);
    // ---- the decision (every wave for itself: the same bits)
    bool ok = false;
    if (p_acc >= 0) {
        double la = be * ((po1[0] + po1[1]) - (po[0] + po[1])) + lq;     // heated chains of MC3: posterior^beta; beta = 1 is exact
        if (jac_root_acc) la += (double)jac_root_acc * (po1[2] - po[2]);   // +1: jf(y) / jf(x); -1 (experiments): the reciprocal
        double ua, ub;
        philox_block(mh_rng(seed, M.chain0 + b, step_acc), 0xFFFFFFFFu, ua, ub);
        ok = (la >= 0) || (ua < exp(la));
        if (tid == 0) {
            const int64_t i = b * M.n_prop + p_acc;
            atomicAdd(&M.tried[i], 1);                      // (no value returned: nothing waits for it)
            if (ok) atomicAdd(&M.acc[i], 1);
            if (trace_alpha) trace_alpha[b] = la;
            if (trace_accept) trace_accept[b] = ok ? 1 : 0;
        }
    }
    MHS_T(2);
    // ---- the current state after the decision: heights and rates, the kept summands of the ln prior, the distances and z of the
    // incremental evaluation -- one batch of loads, then LDS (and back to global memory what an accepted proposal changed)
    double sc[5], pc[3];
#pragma unroll
    for (int i = 0; i < 5; ++i) sc[i] = ok ? scB[i] : scA[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) pc[i] = ok ? pcB[i] : pcA[i];
    // The per-node summands of the birth-death and the clock block of the CURRENT state are kept between launches (psum: two
    // buffers per block and chain, psel says which is current): a proposal that moves a few heights or rates re-evaluates those
    // nodes' summands only -- the same function results as a full evaluation, added in the same order: the same bits.  A launch
    // that evaluates a block writes it to the other buffer; accepting the proposal flips the block's bit.  The first launch of a
    // run (no pending proposal) evaluates the current state's summands in full.
    int sel = sel_pf.x;
    if (ok) sel ^= (sel_pf.y >> 1) & 3;
    const bool take = ok && incr && I.mode != 0;           // the accepted proposal's distances and z become the current ones
    double zv[KZ] = {0.0, 0.0, 0.0, 0.0};
    {
        const double* Hsrc = (ok ? M.H1 : M.H) + b * M.ld;
        const double* Rsrc = (ok ? M.R1 : M.R) + b * M.ld;
        const double* s_bd = M.psum + ((size_t)b * 4 + (size_t)(sel & 1)) * NS;
        const double* s_cl = M.psum + ((size_t)b * 4 + 2 + (size_t)((sel >> 1) & 1)) * NS;
        const double* xsrc = incr ? ((take ? X1 : I.X0) + b * ldx) : nullptr;
        const bool from_tiles = take && I.mode == 2;       // z' of a dense proposal: the row-split kernel's tiles
        const double* zsrc = !incr ? nullptr : from_tiles ? I.zt + ((b >> 4) * I.nr) * 16 + (b & 15) : (take ? I.zprop : I.zcur) + b * I.NPz;
        const bool kept = cached && !cache_init;
        double hv[KM], rv[KM], bv[KM], cv[KM], xv[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            const int w = tid + NT * k;
            hv[k] = (w < n) ? Hsrc[w] : 0.0;
            rv[k] = (w < n) ? Rsrc[w] : 0.0;
            bv[k] = (kept && w < NS) ? s_bd[w] : 0.0;
            cv[k] = (kept && w < NS) ? s_cl[w] : 0.0;
            xv[k] = (incr && w < n_dim) ? xsrc[w] : 0.0;
        }
        if (incr) {
#pragma unroll
            for (int k = 0; k < KZ; ++k) {
                const int i = tid + NT * k;
                if (i < I.NPz) zv[k] = from_tiles ? ((i < I.nr) ? zsrc[(int64_t)i * 16] : 0.0) : zsrc[i];
            }
        }
        MHS_W(3);
#pragma unroll
        for (int k = 0; k < KM; ++k) {                     // LDS first: a store to global memory between two of these would make the
            const int w = tid + NT * k;                    // next one wait for that store's acknowledgement, not just for its load
            if (w < n) {
                par[w] = ix_p[k];
                Hc[w] = hv[k];
                Rc[w] = rv[k];
            }
            if (dist && w < n_dim) {
                snode[w] = ix_a[k];
                spar[w] = ix_q[k];
            }
            if (kept && w < NS) {
                tb[w] = bv[k];
                tc[w] = cv[k];
            }
            if (inc_cols && w < n_dim) incsh.dl[w] = xv[k];   // (the current distances, until the proposal's are known)
        }
        if (ok) {
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                const int w = tid + NT * k;
                if (w < n) {
                    M.H[b * M.ld + w] = hv[k];
                    M.R[b * M.ld + w] = rv[k];
                }
                if (take && w < n_dim) I.X0[b * ldx + w] = xv[k];
            }
        }
        if (take) {
#pragma unroll
            for (int k = 0; k < KZ; ++k) {
                const int i = tid + NT * k;
                if (i < I.NPz) I.zcur[b * I.NPz + i] = zv[k];
            }
        }
        if (accumulate_now) {
            double s1[KM], s2[KM];
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                const int w = tid + NT * k;
                s1[k] = (w < n) ? M.age_sum[b * n + w] : 0.0;
                s2[k] = (w < n) ? M.age_sq[b * n + w] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                const int w = tid + NT * k;
                if (w < n) {
                    const double a = sc[2] * hv[k];
                    M.age_sum[b * n + w] = s1[k] + a;
                    M.age_sq[b * n + w] = s2[k] + a * a;
                }
            }
        }
    }
    if (ok && tid == 0) {                                  // (one lane, plain stores: a select by thread index becomes an indexed private array)
#pragma unroll
        for (int i = 0; i < 5; ++i) M.sc[i * B + b] = sc[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            M.post[i * B + b] = po1[i];
            M.pcomp[b * 3 + i] = pc[i];
        }
    }
    if (p_prop < 0) {
        if (cached && tid == 0) reinterpret_cast<int2*>(M.psel)[b] = make_int2(sel, 0);
        return;
    }
    __syncthreads();                                       // the current state is in LDS
    MHS_T(4);
    ClockCache cc{__builtin_nan(""), 0.0, 0.0, 0.0};
    if (cached && cache_init) {                            // the summands of the current state, in full, into buffer 0 of both blocks
        const bool near0 = prior_bd_near(sc[0], sc[1]);
        prior_clock_scalars(sc[4], cc);
        double* s_bd = M.psum + (size_t)b * 4 * NS;
        double* s_cl = s_bd + 2 * (size_t)NS;
        for (int it = wave; it < NIT; it += MHW) {
            const int v = 1 + lane + 64 * it;
            if (v < n) {
                const double t1 = prior_bd_term(P, v, near0, sc[0], sc[1], Hc);
                const double t2 = prior_clock_term(P, v, sc[4], cc.lg_k, cc.log_t, Hc, Rc);
                tb[it * 64 + lane] = t1;
                tc[it * 64 + lane] = t2;
                s_bd[it * 64 + lane] = t1;
                s_cl[it * 64 + lane] = t2;
            }
        }
    }
    double lnqj = 0.0;
    PropApply* Ap = reinterpret_cast<PropApply*>(bc + 16);  // the per-node transform of the proposal, wave 0 -> all
    if (wave == 0) {
        const double t = M.tune[b * M.n_prop + p_prop];
        const double* dw = M.draws + ((size_t)draw_slot * B + b) * 5;     // the state-independent draws of this step (k_mh_draws)
        const StepDraws dr{dw[0], dw[1], dw[2], dw[3], dw[4]};
        double scp[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) scp[i] = sc[i];
        PropApply A0;
        lnqj = mh_propose_params(M, row_prop, t, dr, lane, scp, Hc, Rc, A0);       // the scalar part: one wave's work
        if (lane == 0) {
            *Ap = A0;
#pragma unroll
            for (int i = 0; i < 5; ++i) bc[i] = scp[i];
            // which blocks of the ln prior the proposed SCALARS move (the heights and rates are compared below); bit 3: the rate
            // variance itself -- every summand of the clock block
            bc[6] = (double)((scp[2] != sc[2] ? 1 : 0) | ((scp[0] != sc[0] || scp[1] != sc[1]) ? 2 : 0) | ((scp[3] != sc[3] || scp[4] != sc[4]) ? 4 : 0) |
                             (!(scp[4] == sc[4]) ? 8 : 0));
            bc[9] = 0.0;
            bc[10] = 0.0;
        }
    } else if (!(cc.va == sc[4])) {
        prior_clock_scalars(sc[4], cc);                    // while wave 0 proposes: most proposals leave the rate variance alone
    }
    __syncthreads();                                       // the transform is in LDS
    MHS_T(5);
    {
        const PropApply A = *Ap;
        bool mH = false, mR = false;
#pragma unroll
        for (int k = 0; k < KM; ++k) {                     // threads = nodes: the proposed state, and whether it differs
            const int w = tid + NT * k;
            if (w < n) {
                double h, r;
                mh_propose_node(M, A, w, Hc, Rc, h, r);
                Hs[w] = h;
                Rs[w] = r;
                mH = mH || (h != Hc[w]);                   // NaN != NaN: re-evaluated
                mR = mR || (r != Rc[w]);
                M.H1[b * M.ld + w] = h;
                M.R1[b * M.ld + w] = r;
            }
        }
        const bool wH = __builtin_amdgcn_ballot_w64(mH) != 0, wR = __builtin_amdgcn_ballot_w64(mR) != 0;
        if (lane == 0) {
            if (wH) bc[9] = 1.0;
            if (wR) bc[10] = 1.0;
        }
    }
    __syncthreads();                                       // the proposed state is in LDS
    MHS_T(6);
    // only the blocks of the ln prior whose inputs the proposal moved are evaluated again
    const bool dH = bc[9] != 0.0, dR = bc[10] != 0.0;
    const int scf = (int)bc[6];
    const int flags = ((dH || (scf & 1)) ? 1 : 0) | ((dH || (scf & 2)) ? 2 : 0) | ((dR || (scf & 4) || (dH && P.clock_model >= 2)) ? 4 : 0);
    double scn[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) scn[i] = bc[i];
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) M.sc1[i * B + b] = scn[i];
    }
    if (tid == 0) {
        if (!prior_inline) M.pflags[b] = flags;
        if (cached) reinterpret_cast<int2*>(M.psel)[b] = make_int2(sel, flags);
    }
    // The incremental evaluation of a sparse proposal (k_mh_inc.hip): which distances moved, in row order -- every wave lists its
    // quarter of the rows, the four lists read one after the other are the row order
    if (dist) {
        // The distances of the proposed state, so that the likelihood launch takes them as a plain vector: the proposal is in LDS
        // here, while the row-split kernel's tree staging gathers it from global memory a chain at a time (25.7 against 19.3 us at
        // 1023 slots).  The arithmetic of that staging (k_split.hip) and of load_tree: ((h_parent - h_node) * rate [+ the second
        // root branch]) * (tH * rMu); ln jacobianRootBranch from slot 0 (app/Probability.hs:201-207, 394, 409).
        const double s = scn[2] * scn[3];
        const int rr = T.root_right;
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            const int j = tid + NT * k;
            if (j < n_dim) {
                const int a = snode[j], pa = spar[j];
                double d = (Hs[pa] - Hs[a]) * Rs[a];
                if (j == 0) {
                    d = d + (Hs[0] - Hs[rr]) * Rs[rr];
                    d = d * s;
                    M.post1[2 * B + b] = log(1.0 / d);
                } else {
                    d = d * s;
                }
                X1[b * ldx + j] = d;
                if (inc_cols) incsh.dl[j] = d - incsh.dl[j];                // delta = x1 - x0 (the same thread left x0 there)
            }
        }
        if (incr && I.prop_mode == 0 && tid == 0) M.post1[B + b] = ok ? po1[1] : po[1];   // a proposal that leaves the distances alone
    }
    if (!prior_inline) {                                   // the ln prior is evaluated beside the likelihood (mh_prior_role.hpp)
        if (tid == 0) M.lnqj[b] = lnqj;
        return;
    }
    if (inc_cols) {
        __syncthreads();                                   // the deltas are in LDS
        const int NPq = I.NPz >> 2;                        // rows per wave: a multiple of 16
        int c = 0;
        for (int j0 = wave * NPq; j0 < (wave + 1) * NPq; j0 += 64) {
            const int j = j0 + lane;
            const bool mv = (j < (wave + 1) * NPq) && (j < n_dim) && incsh.dl[j] != 0.0;   // NaN != 0: kept, and the NaN then reaches q
            const uint64_t mk = __builtin_amdgcn_ballot_w64(mv);
            if (mv) incsh.list[wave * NPq + c + __builtin_popcountll(mk & ((1ull << lane) - 1ull))] = j;
            c += __builtin_popcountll(mk);
        }
        if (lane == 0) incsh.cnt4[wave] = c;
    }
    {                                                      // every wave takes its share of the 64-node iterations (wave 0 as well:
        const int wi = wave;                               // it would only wait)
        if (flags & 2) {
            // a summand depends on its node's and the parent's height, and on the two rates: every one when those moved -- or near
            // the critical case, where a summand is composed along a path to a tip
            const bool near = prior_bd_near(scn[0], scn[1]);
            const bool all = !cached || (scf & 2) || near;
            for (int it = wi; it < NIT; it += MHW) {
                const int v = 1 + lane + 64 * it;
                if (v < n) {
                    const int pv = par[v];
                    if (all || Hs[v] != Hc[v] || Hs[pv] != Hc[pv]) tb[it * 64 + lane] = prior_bd_term(P, v, near, scn[0], scn[1], Hs);
                }
            }
        }
        if (flags & 4) {
            // uncorrelated models: a summand depends on its node's rate and on the rate variance; the others also on the branch's duration
            const bool all = !cached || (scf & 8);
            const bool durations = P.clock_model >= 2;
            if (!(cc.va == scn[4])) prior_clock_scalars(scn[4], cc);
            for (int it = wi; it < NIT; it += MHW) {
                const int v = 1 + lane + 64 * it;
                if (v < n) {
                    bool again = all || Rs[v] != Rc[v];
                    if (durations && !again) {
                        const int pv = par[v];
                        again = Hs[v] != Hc[v] || Hs[pv] != Hc[pv];
                    }
                    if (again) tc[it * 64 + lane] = prior_clock_term(P, v, scn[4], cc.lg_k, cc.log_t, Hs, Rs);
                }
            }
            if (wi == 1 && lane == 0) bc[8] = cc.hyper;
        }
        if ((flags & 1) && wi == MHW - 1) {
            const double c0 = prior_nodes_wave(P, lane, scn[2], Hs);
            if (lane == 0) bc[7] = c0;
        }
    }
    __syncthreads();                                       // the summands and the lists of moved distances are in LDS
    MHS_T(7);
    if (inc_cols) {
        // z' = z + sum_j delta_j W[:, j] over the moved rows j in row order, four columns in flight; ln likelihood = c - 1/2 (logdet + |z'|^2)
        const int NPq = I.NPz >> 2;
        const int c0 = incsh.cnt4[0], c1 = c0 + incsh.cnt4[1], c2 = c1 + incsh.cnt4[2], cnt = c2 + incsh.cnt4[3];
        auto entry = [&](int m) { return (m < c0) ? incsh.list[m] : (m < c1) ? incsh.list[NPq + m - c0] : (m < c2) ? incsh.list[2 * NPq + m - c1] : incsh.list[3 * NPq + m - c2]; };
        const int NP = I.NPz;
        for (int m0 = 0; m0 < cnt; m0 += 4) {
            double w[4][KZ], d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int m = (m0 + u < cnt) ? m0 + u : cnt - 1;
                const int j = entry(m);
                d[u] = (m0 + u < cnt) ? incsh.dl[j] : 0.0;
                const double* wc = V.Wc + (size_t)j * NP;
#pragma unroll
                for (int k = 0; k < KZ; ++k) w[u][k] = (tid + NT * k < NP) ? wc[tid + NT * k] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < KZ; ++k) zv[k] = fma(d[u], w[u][k], zv[k]);
        }
        double sq = 0.0;
        double* zo = I.zprop + b * NP;
#pragma unroll
        for (int k = 0; k < KZ; ++k) {
            if (tid + NT * k < NP) zo[tid + NT * k] = zv[k];
            sq = fma(zv[k], zv[k], sq);
        }
        sq = inc_wave_sum(sq);
        if (lane == 0) incsh.red[wave] = sq;
    }
    if (cached && (flags & 6)) {                           // the proposal's blocks, whole, to the buffers that are not the current ones
        double* s_bd = M.psum + ((size_t)b * 4 + (size_t)((sel & 1) ^ 1)) * NS;
        double* s_cl = M.psum + ((size_t)b * 4 + 2 + (size_t)(((sel >> 1) & 1) ^ 1)) * NS;
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            const int i = tid + NT * k;
            if ((flags & 2) && i < NS) s_bd[i] = tb[i];
            if ((flags & 4) && i < NS) s_cl[i] = tc[i];
        }
    }
    if (inc_cols) {
        __syncthreads();                                   // the four partial |z'|^2
        if (tid == 0) {
            const double q = ((incsh.red[0] + incsh.red[1]) + incsh.red[2]) + incsh.red[3];
            M.post1[B + b] = V.c + (-0.5) * (V.logdet + q);                  // app/Probability.hs:169
        }
    }
    MHS_T(8);
    if (wave != 0) return;
    const double c0p = (flags & 1) ? bc[7] : pc[0];
    double c1p = pc[1], c2p = pc[2];
    if (flags & 2) {
        double bd = 0.0;
        for (int it = 0; it < NIT; ++it)
            if (1 + lane + 64 * it < n) bd += tb[it * 64 + lane];
        c1p = prior_bd_finish(pr_wave_sum(bd), scn[0], scn[1]);
    }
    if (flags & 4) {
        double clock = 0.0;
        for (int it = 0; it < NIT; ++it)
            if (1 + lane + 64 * it < n) clock += tc[it * 64 + lane];
        c2p = prior_clock_finish(P, pr_wave_sum(clock), scn[3], scn[4], bc[8]);
    }
    if (lane == 0) {
        M.pcomp1[b * 3 + 0] = c0p;
        M.pcomp1[b * 3 + 1] = c1p;
        M.pcomp1[b * 3 + 2] = c2p;
        M.lnqj[b] = lnqj;
        M.post1[b] = c0p + c1p + c2p;
    }
#ifdef MCD_MHSTEP_STAMP
    __builtin_amdgcn_s_waitcnt(0);
    MHS_T(9);
    if (b == 0 && tid == 0) {
        const int k = row_prop.kind & 31;
        for (int i = 0; i < 9; ++i) g_mhs_acc[k * 10 + i] += mhs[i + 1] - mhs[i];
        g_mhs_cnt[k] += 1;
    }
#endif
}

// The state-independent draws (gamma multipliers with ratio and logarithm, the uniforms) of up to 64 consecutive steps: one
// THREAD per (step, chain) instead of one wave per chain, so the transcendental work is not repeated on 64 lanes.
__global__ __launch_bounds__(256) void k_mh_draws(MhDev M, const int32_t* __restrict__ sched, int64_t idx0, int count, uint64_t step0,
                                                  uint64_t seed)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)count * M.batch) return;
    const int j = (int)(i / M.batch);
    const int64_t b = i - (int64_t)j * M.batch;
    const int p = sched[idx0 + j];
    const StepDraws d = mh_step_draws(mh_load_row(M, p), M.tune[b * M.n_prop + p], mh_rng(seed, M.chain0 + b, step0 + (uint64_t)j));
    double* o = M.draws + ((size_t)j * M.batch + b) * 5;
    o[0] = d.u;
    o[1] = d.lnq;
    o[2] = d.logu;
    o[3] = d.U;
    o[4] = d.Uacc;
}

// mcmc's auto tuning [external]: t' = clamp(t exp(2 (rate - optimal(dim))), 1e-5, 1e3); counters reset.
__global__ __launch_bounds__(256) void k_mh_tune(MhDev M)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M.batch * M.n_prop) return;
    const int n_tried = M.tried[i];
    if (n_tried > 0) {
        const int dim = M.dim[i % M.n_prop];
        const double opt = mh_optimal_rate(dim);
        const double r = (double)M.acc[i] / (double)n_tried;
        double t = M.tune[i] * exp(2.0 * (r - opt));
        t = (t < 1e-5) ? 1e-5 : (t > 1e3) ? 1e3 : t;
        M.tune[i] = t;
    }
    M.acc[i] = 0;
    M.tried[i] = 0;
}

hipError_t launch_mh_draws(const MhDev& M, const int32_t* sched, int64_t idx0, int count, uint64_t step0, uint64_t seed, hipStream_t st)
{
    const int64_t n = (int64_t)count * M.batch;
    hipLaunchKernelGGL(k_mh_draws, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, M, sched, idx0, count, step0, seed);
    return hipGetLastError();
}
// does launch_mh_step take the workgroup-per-chain kernel for this handle (MCD_MH_STEP_WG: 1 = for every tree, 0 = never)?
// dynamic LDS of k_mh_step_wg: four state vectors, two blocks of summands, the hand-over words, three index arrays
static size_t mh_step_wg_lds(int n_nodes)
{
    const size_t NS = (size_t)((n_nodes - 1 + 63) / 64) * 64;
    return sizeof(double) * (4 * (size_t)n_nodes + 2 * NS + 48) + sizeof(int) * 3 * (size_t)n_nodes;
}
bool mh_step_wg_active(const MhDev& M, int prior_inline, int min_nodes)
{
    const int force = opt_get(OPT_MH_STEP_WG);            // (mcd_set_option "MCD_MH_STEP_WG": 1 / 0 = for every tree / never)
    const bool wg = force != MCD_OPT_UNSET ? force != 0 : (prior_inline && M.n_nodes > min_nodes);
    return wg && M.n_nodes <= 2048 && mh_step_wg_lds(M.n_nodes) + sizeof(IncShared) <= 144 * 1024;   // (above 64 KiB: allowed at launch)
}

hipError_t launch_mh_step(const MhDev& M, const PriorDev& P, int p_acc, int jac_root_acc, int p_prop, const MhRow& r, int draw_slot,
                          uint64_t step_acc, uint64_t seed, int accumulate_now, double* trace_alpha, int8_t* trace_accept, int prior_inline,
                          const TreeDev* T, int n_dim, double* X1, int64_t ldx, hipStream_t st, const MhInc* inc, const MvnDev* V, int summands_init)
{
    if (summands_init < 0) summands_init = p_acc < 0 ? 1 : 0;
    const PropRow row_{r.kind, r.node, r.n1, r.n2, r.jac_root, r.p0, r.p1};
    if (mh_step_wg_active(M, prior_inline, X1 != nullptr ? 0 : 320)) {      // (a caller that wants the distances has asked mh_step_wg_active itself)
        const bool dist = T != nullptr && X1 != nullptr;
        const TreeDev Tv = dist ? *T : TreeDev{};
        double* Xv = dist ? X1 : (double*)nullptr;
        const MhInc Iv = (dist && inc && V) ? *inc : MhInc{};
        const MvnDev Vv = (dist && inc && V) ? *V : MvnDev{};
        auto go = [&](auto km) -> hipError_t {
            constexpr int KM = decltype(km)::value;
            const size_t lds = mh_step_wg_lds(M.n_nodes);
            if (lds + sizeof(IncShared) > 64 * 1024) {       // more than 64 KiB of LDS has to be allowed once per device
                static std::atomic<unsigned long long> allowed{0};
                int dev = 0;
                if (hipError_t e = hipGetDevice(&dev)) return e;
                if (dev < 0 || dev >= 64) return hipErrorInvalidValue;
                if (!((allowed.load(std::memory_order_acquire) >> dev) & 1ull)) {
                    if (hipError_t e = hipFuncSetAttribute((const void*)k_mh_step_wg<KM>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) return e;
                    allowed.fetch_or(1ull << dev, std::memory_order_release);
                }
            }
            hipLaunchKernelGGL(k_mh_step_wg<KM>, dim3((unsigned)M.batch), dim3(64 * MHW), lds, st, M, P, p_acc, jac_root_acc, p_prop, row_, draw_slot,
                               step_acc, seed, accumulate_now, trace_alpha, trace_accept, prior_inline, Tv, n_dim, Xv, ldx, Iv, Vv, summands_init);
            return hipGetLastError();
        };
        if (dist && n_dim > M.n_nodes) return hipErrorInvalidValue;
        if (M.n_nodes <= 256 * 3) return go(std::integral_constant<int, 3>{});
        if (M.n_nodes <= 256 * 5) return go(std::integral_constant<int, 5>{});
        return go(std::integral_constant<int, 8>{});
    }
    if (X1 != nullptr) return hipErrorInvalidValue;        // (the caller asked for distances: only the workgroup kernel writes them)
    const size_t per_wave = sizeof(double) * 4 * (size_t)M.n_nodes;
    int wpb = 4;
    while (wpb > 1 && per_wave * wpb > 60 * 1024) wpb >>= 1;
    if (per_wave * wpb > 64 * 1024) return hipErrorInvalidValue;
    const PropRow row{r.kind, r.node, r.n1, r.n2, r.jac_root, r.p0, r.p1};
    hipLaunchKernelGGL(k_mh_step, dim3((unsigned)((M.batch + wpb - 1) / wpb)), dim3(64 * wpb), per_wave * wpb, st, M, P, p_acc, jac_root_acc, p_prop,
                       row, draw_slot, step_acc, seed, accumulate_now, trace_alpha, trace_accept, prior_inline);
    return hipGetLastError();
}
hipError_t launch_mh_tune(const MhDev& M, hipStream_t st)
{
    const int64_t n = M.batch * M.n_prop;
    hipLaunchKernelGGL(k_mh_tune, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, M);
    return hipGetLastError();
}

}  // namespace mcd

#ifdef MCD_MHSTEP_STAMP
extern "C" int mcd_mhstep_debug_stamps(unsigned long long* acc, unsigned long long* cnt)
{
    if (hipMemcpyFromSymbol(acc, HIP_SYMBOL(mcd::g_mhs_acc), sizeof(unsigned long long) * 320)) return 1;
    return (int)hipMemcpyFromSymbol(cnt, HIP_SYMBOL(mcd::g_mhs_cnt), sizeof(unsigned long long) * 32);
}
#endif
