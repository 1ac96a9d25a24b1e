// k_mh_chain.hip -- a whole Metropolis-Hastings-Green schedule in ONE launch for trees of at most 64 nodes (gfx950).
//
// One wave owns one chain for the complete schedule (n_iter iterations x steps_per_iter proposals): the chain's state
// (node heights, branch rates, the five scalars, ln prior / ln likelihood / ln jacobianRootBranch), its tuning
// parameters and its acceptance counters live in LDS and registers and are written back once at the end.  The
// row-scaled Cholesky factor of Sigma (N <= 62, at most 31 KB) is staged ONCE per workgroup in LDS and shared by the
// workgroup's waves.  Per step, inside the wave and without any workgroup barrier:
//     propose (mh_device.hpp)  ->  ln prior (prior_device.hpp)  ->  distances + forward solve + dot  ->  accept.
// lanes = nodes for the state and the prior, lanes = rows of the factor for the solve: z_j is broadcast with
// v_readlane and column j of the factor is one conflict-free ds_read_b64 (address j * 64 + lane).
//
// The arithmetic is the one of the two-launch path of larger trees (k_mh.hip with prior_device.hpp, k_tree_logpdf.hip at R = 1): same
// proposal code, same prior code, the same fma order in the column sweep and the same reduction tree, so a chain
// advanced by this kernel is bit-identical to the same chain advanced by that path
// (tests/test_gpu_mh.py::test_chain_kernel_equals_per_phase_kernels).
//
// Reference: the loop this replaces is `mhg`'s iteration of `mcmc` [external] driven from app/Main.hs:460-479 with
// the cycle of app/Definitions.hs:256-278; likelihood app/Probability.hs:166-173, 195-207; jacobianRootBranch :393-410.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "mh_device.hpp"
#include "options.h"
#include "prior_device.hpp"

namespace mcd {

// LW (batches below 1024 chains: SIMDs to spare): a second wave per chain evaluates the ln likelihood of the proposed state --
// distances, forward solve, dot: the same instructions on the same numbers -- while the chain's wave evaluates the ln prior: both depend
// on the proposal only.  The two talk through four LDS words (request, reply, |z|^2, the root slot's distance); the second wave keeps
// no state.  The same bits as the one-wave kernel (tests/test_gpu_mh.py::test_chain_kernel_with_a_likelihood_wave).
struct MhcWords {
    int req;                       // chain wave: count of likelihood requests so far (-1: the schedule is over)
    int resp;                      // likelihood wave: the request it has answered
    double s1;                     // chain wave: tH * rMu of the proposal
    double q;                      // likelihood wave: |z|^2
    double d0;                     // ... and the distance of slot 0 (ln jacobianRootBranch = ln (1 / d0))
    double pad[4];
};
static_assert(sizeof(MhcWords) == 64, "eight doubles of LDS");

template <int WPB, bool LW>
__global__ __launch_bounds__(64 * WPB * (LW ? 2 : 1)) void k_mh_chain(MhDev M, MvnDev V, TreeDev T, PriorDev P, const double* __restrict__ Fp,
                                                       const int32_t* __restrict__ sched, int64_t n_steps, int32_t S,
                                                       int accumulate, uint64_t step0, uint64_t seed,
                                                       double* __restrict__ trace_alpha, int8_t* __restrict__ trace_accept)
{
    extern __shared__ double lds[];
    const int n = V.n, nn = M.n_nodes, NP = M.n_prop;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* Fs = lds;                                   // [n][64]: Fs[j * 64 + i] = L_ij / L_ii (i > j), 0 otherwise
    static_assert(!LW || WPB == 1, "the likelihood wave comes with one chain per workgroup");
    for (int i = threadIdx.x; i < n * 64; i += (int)blockDim.x) Fs[i] = Fp[i];
    const size_t per_wave = 4 * 64 + (size_t)NP + (size_t)NP;          // doubles: Hc Rc Hp Rp | tune | (acc, tried) as int32 pairs
    MhcWords* words = reinterpret_cast<MhcWords*>(lds + (size_t)n * 64 + (size_t)WPB * per_wave);
    if (LW && threadIdx.x == 0) {
        words->req = 0;
        words->resp = 0;
    }
    PriorDev Pl = P;                                    // the calibration and constraint tables in LDS (prior_device.hpp)
    if (prior_node_tables_doubles(P.n_cal, P.n_con) > 0)
        prior_stage_node_tables(Pl, P, reinterpret_cast<double*>(words) + 8, (int)threadIdx.x, (int)blockDim.x);
    __syncthreads();                                    // the only workgroup barrier; waves are independent afterwards
    const int cw = LW ? 0 : wave;                       // the chain's index in the workgroup
    const int64_t b = (int64_t)blockIdx.x * WPB + cw;
    if (b >= M.batch) return;
    double* Hc = lds + (size_t)n * 64 + (size_t)cw * per_wave;
    double* Rc = Hc + 64;
    double* Hp = Rc + 64;
    double* Rp = Hp + 64;
    double* tune = Rp + 64;
    int32_t* acc = reinterpret_cast<int32_t*>(tune + NP);
    int32_t* tried = acc + NP;
    const int64_t B = M.batch;
    // row `lane` of the solve: mean, 1 / L_ii, the node whose branch feeds this distance slot and that node's parent
    const double mu_l = V.mu[lane], iv_l = V.invdiag[lane];
    const int slot = T.slot_node[lane];
    const int slot_par = (slot >= 0) ? T.parent[slot] : 0;
    const int rr = T.root_right;
    // ln likelihood of the state in (Hx, Rx) with tH * rMu = s_: |z|^2 and the root slot's distance
    // likelihoodFunctionWrapper: distances = (tH * rMu) * sumFirstTwo (times * rates)      (app/Probability.hs:195-207)
    auto solve = [&](const double* Hx, const double* Rx, double s_, double& dist0) -> double {
        double dist = 0.0;
        if (slot >= 0) {
            dist = (Hx[slot_par] - Hx[slot]) * Rx[slot];
            if (lane == 0) dist = dist + (Hx[0] - Hx[rr]) * Rx[rr];
            dist = dist * s_;
        }
        dist0 = mh_readlane64(dist, 0);
        double d = (dist - mu_l) * iv_l;
        // column sweep of L z = x - mu, row-scaled.  Eight columns per round: their LDS reads are issued together,
        // then the dependent readlane -> fma chain runs; columns beyond n multiply a zero (z_j = 0 there: exact)
        for (int j0 = 0; j0 < n; j0 += 8) {
            double f[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) f[u] = (j0 + u < n) ? Fs[(j0 + u) * 64 + lane] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double zj = mh_readlane64(d, j0 + u);
                d = fma(-f[u], zj, d);
            }
        }
        return pr_wave_sum(fma(d, d, 0.0));
    };
    if (LW && wave == 1) {
        // ---- the likelihood wave
        lds_vint_t* w_req = lds_vint(&words->req);          // (mh_device.hpp: the words typed as LDS, fences that wait for LDS only)
        int last = 0;
        while (true) {
            int v = *w_req;
            while (v == last) {
                __builtin_amdgcn_s_sleep(1);
                v = *w_req;
            }
            lds_acquire_fence();
            if (v < 0) break;
            double d0;
            const double q = solve(Hp, Rp, words->s1, d0);
            if (lane == 0) {
                words->q = q;
                words->d0 = d0;
            }
            lds_publish_fence();
            *lds_vint(&words->resp) = v;
            last = v;
        }
        return;
    }
    int n_req = 0;                                      // likelihood requests so far (LW)
    Hc[lane] = (lane < nn) ? M.H[b * M.ld + lane] : 0.0;
    Rc[lane] = (lane < nn) ? M.R[b * M.ld + lane] : 0.0;
    Hp[lane] = 0.0;                                     // lanes beyond the tree stay equal in both copies
    Rp[lane] = 0.0;
    for (int i = lane; i < NP; i += 64) {
        tune[i] = M.tune[b * NP + i];
        acc[i] = M.acc[b * NP + i];
        tried[i] = M.tried[b * NP + i];
    }
    double sc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) sc[i] = M.sc[i * B + b];
    double ll = M.post[B + b], lj = M.post[2 * B + b];
    // the three blocks of the ln prior of the current state; a step re-evaluates only the blocks whose inputs moved
    __builtin_amdgcn_wave_barrier();
    double c0 = prior_nodes_wave(Pl, lane, sc[2], Hc);
    double c1 = prior_bd_wave(P, lane, sc[0], sc[1], Hc);
    ClockCache cc{__builtin_nan(""), 0.0, 0.0, 0.0};                    // variance-only pieces of the clock block, current state
    double c2 = prior_clock_wave(P, lane, sc[3], sc[4], Hc, Rc, &cc);
    double lp = c0 + c1 + c2;
    double age_s = 0.0, age_q = 0.0;
    const double beta = M.beta[b];
#ifdef MCD_MH_STAMP
    uint64_t tk[6] = {0, 0, 0, 0, 0, 0};
#define MH_TICK(i)                                        \
    {                                                     \
        const uint64_t now_ = __builtin_readcyclecounter(); \
        tk[i] += now_ - t_last;                           \
        t_last = now_;                                    \
    }
    uint64_t t_last = __builtin_readcyclecounter();
#else
#define MH_TICK(i)
#endif
    __builtin_amdgcn_wave_barrier();
    int p = sched[0];
    const int vz = mh_vzero();                              // (mh_device.hpp: what travels ahead is loaded by vector loads)
    int p_next = sched[(n_steps > 1 ? 1 : 0) + vz];         // (the schedule's entries two steps ahead: a row's loads need its index)
    PropRow row = mh_load_row(M, p);
    StepDraws pre{1.0, 0.0, 0.0, 0.5, 0.5};                 // lane l: the state-independent draws of step (gs & ~63) + l
    for (int64_t gs = 0; gs < n_steps; ++gs) {
        const int p_next2 = sched[((gs + 2 < n_steps) ? gs + 2 : gs) + vz];
        const PropRow row_next = mh_load_row_ahead(M, p_next);      // the next step's row travels while this step computes
        if ((gs & 63) == 0) {
            // 64 consecutive steps at once, one step per lane: the random draws that depend only on the proposal row and its
            // tuning parameter (gamma multipliers with their ratio and logarithm, the uniforms of the truncated normals
            // and of the acceptance) cost one evaluation per 64 steps instead of one per step
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
        MH_TICK(0)
        const double lnqj = mh_propose_wave(M, row, tune[p], dr, lane, sc1, Hc, Rc, Hp, Rp);
        __builtin_amdgcn_wave_barrier();
        MH_TICK(1)
        const bool dH = __builtin_amdgcn_ballot_w64(Hp[lane] != Hc[lane]) != 0;     // NaN != NaN: re-evaluated
        const bool dR = __builtin_amdgcn_ballot_w64(Rp[lane] != Rc[lane]) != 0;
        const bool moves = dH || dR || sc1[2] != sc[2] || sc1[3] != sc[3];         // birth, death and rVar do not enter the likelihood
        if (LW && moves) {                                                 // the likelihood wave starts on the proposed state now
            n_req += 1;
            if (lane == 0) words->s1 = sc1[2] * sc1[3];
            lds_publish_fence();
            *lds_vint(&words->req) = n_req;
        }
        ClockCache ccp = cc;                                               // refreshed only if the proposal moved rVar
        const double c0p = (dH || sc1[2] != sc[2]) ? prior_nodes_wave(Pl, lane, sc1[2], Hp) : c0;
        const double c1p = (dH || sc1[0] != sc[0] || sc1[1] != sc[1]) ? prior_bd_wave(P, lane, sc1[0], sc1[1], Hp) : c1;
        const double c2p = (dR || sc1[3] != sc[3] || sc1[4] != sc[4] || (dH && P.clock_model >= 2))
                               ? prior_clock_wave(P, lane, sc1[3], sc1[4], Hp, Rp, &ccp) : c2;
        const double lp1 = c0p + c1p + c2p;
        MH_TICK(2)
        double ll1 = ll, lj1 = lj;
        if (moves) {
            double q, dist0;
            if (LW) {
                lds_vint_t* w_resp = lds_vint(&words->resp);
                while (*w_resp != n_req) __builtin_amdgcn_s_sleep(1);
                lds_acquire_fence();
                q = words->q;
                dist0 = words->d0;
            } else {
                q = solve(Hp, Rp, sc1[2] * sc1[3], dist0);
            }
            lj1 = log(1.0 / dist0);                                        // jacobianRootBranch, :393-410
            ll1 = V.c + (-0.5) * (V.logdet + q);                           // :169
        }
        MH_TICK(3)
        double la = beta * ((lp1 + ll1) - (lp + ll)) + lnqj;           // heated chains of MC3: posterior^beta; beta = 1 is exact
        if (row.jac_root) la += (double)row.jac_root * (lj1 - lj);
        const bool ok = (la >= 0) || (dr.Uacc < exp(la));
        if (ok) {
            Hc[lane] = Hp[lane];
            Rc[lane] = Rp[lane];
#pragma unroll
            for (int i = 0; i < 5; ++i) sc[i] = sc1[i];
            c0 = c0p;
            c1 = c1p;
            c2 = c2p;
            cc = ccp;
            lp = lp1;
            ll = ll1;
            lj = lj1;
        }
        if (lane == 0) {
            tried[p] += 1;
            if (ok) acc[p] += 1;
            if (trace_alpha) trace_alpha[gs * B + b] = la;
            if (trace_accept) trace_accept[gs * B + b] = ok ? 1 : 0;
        }
        __builtin_amdgcn_wave_barrier();
        if (accumulate && (gs + 1) % S == 0) {
            const double a = sc[2] * Hc[lane];
            age_s += a;
            age_q += a * a;
        }
        p = __builtin_amdgcn_readfirstlane(p_next);
        p_next = p_next2;
        row = mh_row_scalar(row_next);
        MH_TICK(4)
    }
    if (LW) {                                           // the schedule is over
        lds_publish_fence();
        *lds_vint(&words->req) = -1;
    }
#ifdef MCD_MH_STAMP
    if (trace_alpha && lane == 0)
        for (int i = 0; i < 5; ++i) trace_alpha[(int64_t)i * B + b] = (double)tk[i];   // cycles: loop head, propose, prior, likelihood, accept
#endif
    if (lane < nn) {
        M.H[b * M.ld + lane] = Hc[lane];
        M.R[b * M.ld + lane] = Rc[lane];
        if (accumulate) {
            M.age_sum[b * nn + lane] += age_s;
            M.age_sq[b * nn + lane] += age_q;
        }
    }
    for (int i = lane; i < NP; i += 64) {
        M.acc[b * NP + i] = acc[i];
        M.tried[b * NP + i] = tried[i];
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
    }
}

size_t mh_chain_lds_bytes(int n, int n_prop, int wpb)
{
    return sizeof(double) * ((size_t)n * 64 + (size_t)wpb * (4 * 64 + 2 * (size_t)n_prop) + 8);       // (+ the hand-over words of the two-wave form; the caller adds the node priors' tables)
}

hipError_t launch_mh_chain(const MhDev& M, const MvnDev& V, const TreeDev& T, const PriorDev& P, const double* Fp,
                           const int32_t* sched, int64_t n_steps, int32_t S, int accumulate, uint64_t step0, uint64_t seed,
                           double* trace_alpha, int8_t* trace_accept, hipStream_t st)
{
    if (n_steps <= 0) return hipSuccess;
    if (M.batch >= 1024) {
        constexpr int WPB = 4;
        const size_t sh = mh_chain_lds_bytes(V.n, M.n_prop, WPB) + sizeof(double) * prior_node_tables_doubles(P.n_cal, P.n_con);
        hipLaunchKernelGGL((k_mh_chain<WPB, false>), dim3((unsigned)((M.batch + WPB - 1) / WPB)), dim3(64 * WPB), sh, st, M, V, T, P, Fp, sched,
                           n_steps, S, accumulate, step0, seed, trace_alpha, trace_accept);
    } else {
        constexpr int WPB = 1;
        const size_t sh = mh_chain_lds_bytes(V.n, M.n_prop, WPB) + sizeof(double) * prior_node_tables_doubles(P.n_cal, P.n_con);
        if (opt_is(OPT_MH_CHAIN_LW, 0))                   // (mcd_set_option "MCD_MH_CHAIN_LW" = 0: one wave per chain; tests, timing)
            hipLaunchKernelGGL((k_mh_chain<WPB, false>), dim3((unsigned)M.batch), dim3(64), sh, st, M, V, T, P, Fp, sched, n_steps, S, accumulate,
                               step0, seed, trace_alpha, trace_accept);
        else
            hipLaunchKernelGGL((k_mh_chain<WPB, true>), dim3((unsigned)M.batch), dim3(128), sh, st, M, V, T, P, Fp, sched, n_steps, S, accumulate,
                               step0, seed, trace_alpha, trace_accept);
    }
    return hipGetLastError();
}

}  // namespace mcd
