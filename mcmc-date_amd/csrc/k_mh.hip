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

#include <atomic>
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
__global__ __launch_bounds__(64 * MHW, 2) void k_mh_step_wg(MhDev M, PriorDev P, int p_acc, int jac_root_acc, int p_prop, PropRow row_prop,
                                                            int draw_slot, uint64_t step_acc, uint64_t seed, int accumulate_now,
                                                            double* __restrict__ trace_alpha, int8_t* __restrict__ trace_accept, int prior_inline,
                                                            TreeDev T, int n_dim, double* __restrict__ X1, int64_t ldx, MhInc I, MvnDev V)
{
    __shared__ IncShared incsh;
    extern __shared__ double sh[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NT = 64 * MHW;
    const int64_t b = blockIdx.x;
    const int n = M.n_nodes;
    const int64_t B = M.batch;
    const int NIT = (n - 1 + 63) >> 6;                     // iterations of v = 1 + lane + 64 it < n
    double* Hc = sh;                                       // current state after the decision
    double* Rc = Hc + n;
    double* Hs = Rc + n;                                   // proposed state
    double* Rs = Hs + n;
    double* tb = Rs + n;                                   // [NIT][64] summands of the birth-death block
    double* tc = tb + NIT * 64;                            // [NIT][64] summands of the clock block
    double* bc = tc + NIT * 64;                            // [48] wave 0 -> all: proposed scalars, flags, the per-node transform; workers -> wave 0: c0, hyper
#ifdef MCD_MHSTEP_STAMP
    unsigned long long mhs[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    MHS_T(0);
    // ---- the decision (every wave for itself: the same bits)
    bool ok = false;
    if (p_acc >= 0) {
        const double lp = M.post[b], ll = M.post[B + b], lj = M.post[2 * B + b];
        const double lp1 = M.post1[b], ll1 = M.post1[B + b], lj1 = M.post1[2 * B + b];
        double la = M.beta[b] * ((lp1 + ll1) - (lp + ll)) + M.lnqj[b];     // heated chains of MC3: posterior^beta; beta = 1 is exact
        if (jac_root_acc) la += (double)jac_root_acc * (lj1 - lj);   // +1: jf(y) / jf(x); -1 (experiments): the reciprocal
        double ua, ub;
        philox_block(mh_rng(seed, M.chain0 + b, step_acc), 0xFFFFFFFFu, ua, ub);
        ok = (la >= 0) || (ua < exp(la));
        if (tid == 0) {
            const int64_t i = b * M.n_prop + p_acc;
            M.tried[i] += 1;
            if (ok) M.acc[i] += 1;
            if (trace_alpha) trace_alpha[b] = la;
            if (trace_accept) trace_accept[b] = ok ? 1 : 0;
        }
    }
    MHS_T(1);
    // the current state after the decision, into LDS (and back to global memory when it changed)
    double sc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) sc[i] = (ok ? M.sc1 : M.sc)[i * B + b];
    {
        const double* Hsrc = (ok ? M.H1 : M.H) + b * M.ld;
        const double* Rsrc = (ok ? M.R1 : M.R) + b * M.ld;
        for (int w = tid; w < n; w += NT) {
            const double h = Hsrc[w], r = Rsrc[w];
            Hc[w] = h;
            Rc[w] = r;
            if (ok) {
                M.H[b * M.ld + w] = h;
                M.R[b * M.ld + w] = r;
            }
            if (accumulate_now) {
                const double a = sc[2] * h;
                M.age_sum[b * n + w] += a;
                M.age_sq[b * n + w] += a * a;
            }
        }
    }
    // The per-node summands of the birth-death and the clock block of the CURRENT state are kept between launches (psum: two
    // buffers per block and chain, psel says which is current): a proposal that moves a few heights or rates re-evaluates those
    // nodes' summands only -- the same function results as a full evaluation, added in the same order: the same bits.  A launch
    // that evaluates a block writes it to the other buffer; accepting the proposal flips the block's bit.  The first launch of a
    // run (no pending proposal) evaluates the current state's summands in full.
    const bool cached = prior_inline && M.psum != nullptr;
    const bool cache_init = p_acc < 0;
    const int NS = NIT * 64;
    int sel = 0;
    if (cached && !cache_init) {
        sel = M.psel[b];
        if (ok) sel ^= (M.pflags[b] >> 1) & 3;             // the accepted proposal's blocks become the current ones
        const double* s_bd = M.psum + ((size_t)b * 4 + (size_t)(sel & 1)) * NS;
        const double* s_cl = M.psum + ((size_t)b * 4 + 2 + (size_t)((sel >> 1) & 1)) * NS;
        for (int i = tid; i < NS; i += NT) {
            tb[i] = s_bd[i];
            tc[i] = s_cl[i];
        }
    }
    double pc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) pc[i] = (ok ? M.pcomp1 : M.pcomp)[b * 3 + i];
    if (ok) {
        if (tid < 5) {
            double mine = sc[0];
#pragma unroll
            for (int i = 1; i < 5; ++i)
                if (tid == i) mine = sc[i];
            M.sc[tid * B + b] = mine;
        }
        if (tid < 3) {
            M.post[tid * B + b] = M.post1[tid * B + b];
            double mine = pc[0];
            if (tid == 1) mine = pc[1];
            if (tid == 2) mine = pc[2];
            M.pcomp[b * 3 + tid] = mine;
        }
    }
    if (ok && I.X0 != nullptr && I.mode != 0) {
        // incremental likelihood (k_mh_inc.hip): the accepted proposal's distances and z = L^-1 (d - mu) become the current ones --
        // z' from where the pending proposal's likelihood step left it.  (X1 is rewritten below by the same threads, in program order.)
        for (int j = tid; j < n_dim; j += NT) I.X0[b * ldx + j] = X1[b * ldx + j];
        double* zc = I.zcur + b * I.NPz;
        if (I.mode == 1) {
            const double* zp = I.zprop + b * I.NPz;
            for (int i = tid; i < I.NPz; i += NT) zc[i] = zp[i];
        } else {
            const double* zt = I.zt + ((b >> 4) * I.nr) * 16 + (b & 15);
            for (int i = tid; i < I.NPz; i += NT) zc[i] = (i < I.nr) ? zt[(int64_t)i * 16] : 0.0;
        }
    }
    if (cached && tid == 0) M.psel[b] = sel;
    if (p_prop < 0) return;
    __syncthreads();                                       // the current state is in LDS
    MHS_T(2);
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
        double sc0[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) sc0[i] = sc[i];
        PropApply A0;
        lnqj = mh_propose_params(M, row_prop, t, dr, lane, sc, Hc, Rc, A0);       // the scalar part: one wave's work
        if (lane == 0) {
            *Ap = A0;
#pragma unroll
            for (int i = 0; i < 5; ++i) bc[i] = sc[i];
            // which blocks of the ln prior the proposed SCALARS move (the heights and rates are compared below)
            bc[6] = (double)((sc[2] != sc0[2] ? 1 : 0) | ((sc[0] != sc0[0] || sc[1] != sc0[1]) ? 2 : 0) | ((sc[3] != sc0[3] || sc[4] != sc0[4]) ? 4 : 0) |
                             (!(sc[4] == sc0[4]) ? 8 : 0));   // (bit 3: the rate variance itself -- every summand of the clock block)
            bc[9] = 0.0;
            bc[10] = 0.0;
        }
    } else if (!(cc.va == sc[4])) {
        prior_clock_scalars(sc[4], cc);                    // while wave 0 proposes: most proposals leave the rate variance alone
    }
    __syncthreads();                                       // the transform is in LDS
    MHS_T(3);
    {
        const PropApply A = *Ap;
        bool mH = false, mR = false;
        for (int w = tid; w < n; w += NT) {                // threads = nodes: the proposed state, and whether it differs
            double h, r;
            mh_propose_node(M, A, w, Hc, Rc, h, r);
            Hs[w] = h;
            Rs[w] = r;
            mH = mH || (h != Hc[w]);                       // NaN != NaN: re-evaluated
            mR = mR || (r != Rc[w]);
        }
        const bool wH = __builtin_amdgcn_ballot_w64(mH) != 0, wR = __builtin_amdgcn_ballot_w64(mR) != 0;
        if (lane == 0) {
            if (wH) bc[9] = 1.0;
            if (wR) bc[10] = 1.0;
        }
    }
    __syncthreads();                                       // the proposed state is in LDS
    MHS_T(4);
    // only the blocks of the ln prior whose inputs the proposal moved are evaluated again
    const bool dH = bc[9] != 0.0, dR = bc[10] != 0.0;
    const int scf = (int)bc[6];
    const int flags = ((dH || (scf & 1)) ? 1 : 0) | ((dH || (scf & 2)) ? 2 : 0) | ((dR || (scf & 4) || (dH && P.clock_model >= 2)) ? 4 : 0);
    double scn[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) scn[i] = bc[i];
    for (int w = tid; w < n; w += NT) {
        M.H1[b * M.ld + w] = Hs[w];
        M.R1[b * M.ld + w] = Rs[w];
    }
    if (tid < 5) {
        double mine = scn[0];
#pragma unroll
        for (int i = 1; i < 5; ++i)
            if (tid == i) mine = scn[i];
        M.sc1[tid * B + b] = mine;
    }
    if (X1 != nullptr) {
        // The distances of the proposed state, so that the likelihood launch takes them as a plain vector: the proposal is in LDS
        // here, while the row-split kernel's tree staging gathers it from global memory a chain at a time (25.7 against 19.3 us at
        // 1023 slots).  The arithmetic of that staging (k_split.hip) and of load_tree: ((h_parent - h_node) * rate [+ the second
        // root branch]) * (tH * rMu); ln jacobianRootBranch from slot 0 (app/Probability.hs:201-207, 394, 409).
        const double s = scn[2] * scn[3];
        const int rr = T.root_right;
        for (int j = tid; j < n_dim; j += NT) {
            const int a = T.slot_node[j], pa = T.slot_parent[j];
            double d = (Hs[pa] - Hs[a]) * Rs[a];
            if (j == 0) {
                d = d + (Hs[0] - Hs[rr]) * Rs[rr];
                d = d * s;
                M.post1[2 * B + b] = log(1.0 / d);
            } else {
                d = d * s;
            }
            X1[b * ldx + j] = d;
        }
        MHS_T(5);
        if (I.X0 != nullptr && I.prop_mode != 2) {
            // incremental likelihood (k_mh_inc.hip): this proposal moves a few distances, or none -- its ln likelihood here, from the
            // current z and columns of L^-1, instead of a likelihood launch
            __syncthreads();                                 // X1 of this chain is written (global, read back by the same workgroup)
            if (I.prop_mode == 1) {
                const double ll1 = mh_inc_ll_block(V, I, X1 + b * ldx, I.X0 + b * ldx, I.zcur + b * I.NPz, I.zprop + b * I.NPz, incsh, tid);
                if (tid == 0) M.post1[B + b] = ll1;
            } else if (tid == 0) {
                M.post1[B + b] = M.post[B + b];
            }
        }
    }
    MHS_T(6);
    if (!prior_inline || cached) {
        if (tid == 0) M.pflags[b] = flags;
    }
    if (!prior_inline) {                                   // the ln prior is evaluated beside the likelihood (mh_prior_role.hpp)
        if (tid == 0) M.lnqj[b] = lnqj;
        return;
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
                    const int pv = P.parent[v];
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
                        const int pv = P.parent[v];
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
    __syncthreads();                                       // the summands are in LDS
    MHS_T(7);
    if (cached && (flags & 6)) {                           // the proposal's blocks, whole, to the buffers that are not the current ones
        double* s_bd = M.psum + ((size_t)b * 4 + (size_t)((sel & 1) ^ 1)) * NS;
        double* s_cl = M.psum + ((size_t)b * 4 + 2 + (size_t)(((sel >> 1) & 1) ^ 1)) * NS;
        if ((flags & 2) && wave != 0)                       // (wave 0 closes the sums meanwhile)
            for (int i = tid - 64; i < NS; i += NT - 64) s_bd[i] = tb[i];
        if ((flags & 4) && wave != 0)
            for (int i = tid - 64; i < NS; i += NT - 64) s_cl[i] = tc[i];
    }
    if (wave != 0) return;
    const double c0p = (flags & 1) ? bc[7] : pc[0];
    double c1p = pc[1], c2p = pc[2];
    if (flags & 2) {
        double bd = 0.0;
        for (int it = 0; it < NIT; ++it)
            if (1 + lane + 64 * it < n) bd += tb[it * 64 + lane];
        c1p = prior_bd_finish(pr_wave_sum(bd), sc[0], sc[1]);
    }
    if (flags & 4) {
        double clock = 0.0;
        for (int it = 0; it < NIT; ++it)
            if (1 + lane + 64 * it < n) clock += tc[it * 64 + lane];
        c2p = prior_clock_finish(P, pr_wave_sum(clock), sc[3], sc[4], bc[8]);
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
    MHS_T(8);
    if (b == 0 && tid == 0) {
        const int k = row_prop.kind & 31;
        for (int i = 0; i < 8; ++i) g_mhs_acc[k * 10 + i] += mhs[i + 1] - mhs[i];
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
bool mh_step_wg_active(const MhDev& M, int prior_inline)
{
    const char* env = getenv("MCD_MH_STEP_WG");
    const bool wg = env ? atoi(env) != 0 : (prior_inline && M.n_nodes > 320);
    const int NIT = (M.n_nodes - 1 + 63) / 64;
    return wg && sizeof(double) * (4 * (size_t)M.n_nodes + 2 * (size_t)NIT * 64 + 48) + sizeof(IncShared) <= 144 * 1024;   // (above 64 KiB: allowed at launch)
}

hipError_t launch_mh_step(const MhDev& M, const PriorDev& P, int p_acc, int jac_root_acc, int p_prop, const MhRow& r, int draw_slot,
                          uint64_t step_acc, uint64_t seed, int accumulate_now, double* trace_alpha, int8_t* trace_accept, int prior_inline,
                          const TreeDev* T, int n_dim, double* X1, int64_t ldx, hipStream_t st, const MhInc* inc, const MvnDev* V)
{
    const PropRow row_{r.kind, r.node, r.n1, r.n2, r.jac_root, r.p0, r.p1};
    if (mh_step_wg_active(M, prior_inline)) {
        const int NIT = (M.n_nodes - 1 + 63) / 64;
        const size_t lds = sizeof(double) * (4 * (size_t)M.n_nodes + 2 * (size_t)NIT * 64 + 48);
        const bool dist = T != nullptr && X1 != nullptr;
        if (lds > 64 * 1024) {                               // trees beyond about 1 300 nodes: more than 64 KiB of LDS has to be allowed once per device
            static std::atomic<unsigned long long> allowed{0};
            int dev = 0;
            if (hipError_t e = hipGetDevice(&dev)) return e;
            if (dev < 0 || dev >= 64 || lds + sizeof(IncShared) > 160 * 1024) return hipErrorInvalidValue;
            if (!((allowed.load(std::memory_order_acquire) >> dev) & 1ull)) {
                if (hipError_t e = hipFuncSetAttribute((const void*)k_mh_step_wg, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) return e;
                allowed.fetch_or(1ull << dev, std::memory_order_release);
            }
        }
        hipLaunchKernelGGL(k_mh_step_wg, dim3((unsigned)M.batch), dim3(64 * MHW), lds, st, M, P, p_acc, jac_root_acc, p_prop, row_, draw_slot,
                           step_acc, seed, accumulate_now, trace_alpha, trace_accept, prior_inline, dist ? *T : TreeDev{}, n_dim,
                           dist ? X1 : (double*)nullptr, ldx, (dist && inc && V) ? *inc : MhInc{}, (dist && inc && V) ? *V : MvnDev{});
        return hipGetLastError();
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
