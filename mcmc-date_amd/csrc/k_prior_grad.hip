// k_prior_grad.hip -- gradient of the log prior with respect to the state (gfx950).  SURVEY.md 8(f) row f3, first part:
// the Hamiltonian target of the reference is prior x likelihood x jacobianRootBranch (app/Hamiltonian.hs:72-92,
// `htargetWith`), differentiated by AD in Haskell; the likelihood part is mcd_tree_grad_batch, this is the prior part.
//
//   d ln priorFunction ht md cb cs bs x / d (birth, death, tH, heights, rMu, rVar, rates)        (app/Probability.hs:127-150)
//
// Mapping: one workgroup per chain, threads = nodes (see the kernel).  Every per-node term of the birth-death block depends on
// (la, mu, h_v, h_parent), every per-node term of the clock block on (r_v, rVar, h_v, h_parent): they are evaluated
// with FORWARD-MODE DUAL NUMBERS carrying four tangents, with exactly the value formulas of prior_device.hpp, so the
// derivative code cannot drift from the value code.  The contribution to the parent's height goes through LDS and is
// gathered by the parent lane from its (at most two) children; the soft node priors (calibrations, constraints,
// braces) are few and are added by lane 0 one after the other (fixed order: bit-reproducible).
// Outside the support (ln prior = -inf or NaN) the gradient is NaN.  In the near-critical regime |la - mu| < 1e-6 the
// reference switches the VALUE to first-order formulas composed along a chain of nodes (BirthDeath.hs:90-118); the
// initial state of every analysis sits there (birth = death = 1, app/Definitions.hs:99-100).  The gradient is then taken
// from the exact formulas at the edge of that regime (la moved to mu +- 1e-6): it differs from the derivative of the
// first-order value by O(1e-6) relative, which a Hamiltonian trajectory does not notice (its accept step uses values).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mvn_kernels.h"
#include "prior_device.hpp"

namespace mcd {

namespace {

struct D4 {
    double v, d[4];
};
__device__ __forceinline__ D4 cst(double c) { return D4{c, {0.0, 0.0, 0.0, 0.0}}; }
__device__ __forceinline__ D4 var(double c, int i)
{
    D4 r = cst(c);
#pragma unroll
    for (int k = 0; k < 4; ++k) r.d[k] = (k == i) ? 1.0 : 0.0;
    return r;
}
__device__ __forceinline__ D4 operator+(const D4& a, const D4& b)
{
    D4 r{a.v + b.v, {}};
#pragma unroll
    for (int k = 0; k < 4; ++k) r.d[k] = a.d[k] + b.d[k];
    return r;
}
__device__ __forceinline__ D4 operator-(const D4& a, const D4& b)
{
    D4 r{a.v - b.v, {}};
#pragma unroll
    for (int k = 0; k < 4; ++k) r.d[k] = a.d[k] - b.d[k];
    return r;
}
__device__ __forceinline__ D4 operator-(const D4& a)
{
    D4 r{-a.v, {}};
#pragma unroll
    for (int k = 0; k < 4; ++k) r.d[k] = -a.d[k];
    return r;
}
__device__ __forceinline__ D4 operator*(const D4& a, const D4& b)
{
    D4 r{a.v * b.v, {}};
#pragma unroll
    for (int k = 0; k < 4; ++k) r.d[k] = a.d[k] * b.v + a.v * b.d[k];
    return r;
}
__device__ __forceinline__ D4 operator/(const D4& a, const D4& b)
{
    const double q = a.v / b.v;
    D4 r{q, {}};
#pragma unroll
    for (int k = 0; k < 4; ++k) r.d[k] = (a.d[k] - q * b.d[k]) / b.v;
    return r;
}
__device__ __forceinline__ D4 dexp(const D4& a)
{
    const double e = exp(a.v);
    D4 r{e, {}};
#pragma unroll
    for (int k = 0; k < 4; ++k) r.d[k] = e * a.d[k];
    return r;
}
__device__ __forceinline__ D4 dlog(const D4& a)
{
    D4 r{log(a.v), {}};
#pragma unroll
    for (int k = 0; k < 4; ++k) r.d[k] = a.d[k] / a.v;
    return r;
}
__device__ __forceinline__ D4 dsqrt(const D4& a)
{
    const double s = sqrt(a.v);
    D4 r{s, {}};
#pragma unroll
    for (int k = 0; k < 4; ++k) r.d[k] = 0.5 * a.d[k] / s;
    return r;
}
// digamma: recurrence up to x >= 10, then the asymptotic series (error < 1e-15)
__device__ __forceinline__ double digamma(double x)
{
    double r = 0.0;
    while (x < 10.0) {
        r -= 1.0 / x;
        x += 1.0;
    }
    const double f = 1.0 / (x * x);
    const double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 + f * (-1.0 / 132.0 + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
    return r + log(x) - 0.5 / x + t;
}
__device__ __forceinline__ D4 dlgamma(const D4& a)
{
    const double psi = digamma(a.v);
    D4 r{lgamma(a.v), {}};
#pragma unroll
    for (int k = 0; k < 4; ++k) r.d[k] = psi * a.d[k];
    return r;
}
// ln gamma(k, t)(x), x > 0 checked by the caller
__device__ __forceinline__ D4 d_ln_gamma_pdf(const D4& k, const D4& t, const D4& x)
{
    return dlog(x) * (k - cst(1.0)) - (x / t) - dlgamma(k) - dlog(t) * k;
}
// logNormal' 1 v x
__device__ __forceinline__ D4 d_ln_lognormal_prime(const D4& v, const D4& x)
{
    const D4 t = -(cst(kLnSqrt2Pi) + dlog(x * dsqrt(v)));
    const D4 a = cst(1.0) / (cst(2.0) * v);
    const D4 b = dlog(x) + cst(0.5) * v;
    return t + (-(a * b * b));
}

}  // namespace

// One WORKGROUP per chain (1 .. 4 waves: a thread per node up to 256 nodes).  With one wave per chain the kernel was a chain of
// four trips through the dual-number exponentials / logarithms for a 255-node tree, and a fifth and sixth through the value
// formulas (23 us for 512 chains, rocprofv3; the largest part of a leapfrog step).  Every per-node quantity goes to LDS; wave 0
// adds them up in the order the one-wave kernel used -- lane by lane over v = lane + 64 it for the tangents, over
// v = 1 + lane + 64 it for the values (prior_bd_wave / prior_clock_wave) -- so the results are the same bits.
__global__ void __launch_bounds__(256, 2) k_prior_grad(PriorDev P, const double* __restrict__ birth, const double* __restrict__ death,
                                                    const double* __restrict__ tH, const double* __restrict__ H,
                                                    const double* __restrict__ rMu, const double* __restrict__ rVar,
                                                    const double* __restrict__ Rt, int64_t lds, int64_t batch,
                                                    double* __restrict__ lp, double* __restrict__ g_birth,
                                                    double* __restrict__ g_death, double* __restrict__ g_tH,
                                                    double* __restrict__ g_H, double* __restrict__ g_rMu,
                                                    double* __restrict__ g_rVar, double* __restrict__ g_R)
{
    extern __shared__ double sh[];
    const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b = blockIdx.x;
    const int n = P.n_nodes;
    double* gh = sh;                                          // [n] d/d h_v
    double* ep = gh + n;                                      // [n] contribution of node v's terms to its parent's height
    double* a_la = ep + n;                                    // [n] per-node tangents wrt birth, death, rate variance
    double* a_mu = a_la + n;
    double* a_va = a_mu + n;
    double* vb = a_va + n;                                    // [n] per-node summands of the VALUE (birth-death, clock)
    double* vc = vb + n;
    double* bc = vc + n;                                      // [2] wave 0 -> all: inside the support?
    const double* h = H + b * lds;
    const double* r = Rt + b * lds;
    const double la = birth[b], mu = death[b], th = tH[b], rm = rMu[b], va = rVar[b];
    const bool near = prior_bd_near(la, mu);
    const double la_d = near ? mu + ((la >= mu) ? 1e-6 : -1e-6) : la;     // see the note on the near-critical regime above
    for (int v = tid; v < n; v += nthr) {
        if (v == 0) {
            gh[0] = 0.0;
            ep[0] = 0.0;
            g_R[b * lds] = 0.0;
            continue;
        }
        const int pv = P.parent[v];
        const int nc = P.n_children[v];
        // ---- birth-death term of node v: tangents (la, mu, h_v, h_parent) -------------------------------------
        const D4 dla = var(la_d, 0), dmu = var(mu, 1), hv = var(h[v], 2), hp = var(h[pv], 3);
        const D4 br = hp - hv;
        D4 term_bd;
        if (br.v <= 0) {
            term_bd = cst(kNegInf);
        } else {
            D4 e0 = cst(0.0);
            if (nc > 0) {
                const D4 xx = dexp(-(dla - dmu) * hv);
                e0 = dmu * (cst(1.0) - xx) / (dla - dmu * xx);
            }
            const D4 d = dla - dmu;                           // computeDE, rho = 1: c = e0
            const D4 x = dexp(-d * br);
            const D4 y = (dmu - e0 * dla) * x;
            const D4 c1 = e0 - cst(1.0);
            const D4 denom = dla * c1 + y;
            const D4 pD = d * d * x / denom / denom;
            term_bd = dlog((nc == 2) ? pD * dla : pD);
        }
        // ---- clock term of node v: tangents (r_v, rVar, h_v, h_parent) ----------------------------------------
        const D4 rate = var(r[v], 0), dva = var(va, 1), hv2 = var(h[v], 2), hp2 = var(h[pv], 3);
        const D4 br2 = hp2 - hv2;
        D4 term_ck;
        if (rate.v <= 0) {
            term_ck = cst(kNegInf);
        } else {
            switch (P.clock_model) {
                case 0: term_ck = d_ln_gamma_pdf(cst(1.0) / dva, dva, rate); break;
                case 1: term_ck = d_ln_lognormal_prime(dva, rate); break;
                case 2: { const D4 v2 = dva / br2; term_ck = d_ln_gamma_pdf(cst(1.0) / v2, v2, rate); } break;
                default: term_ck = d_ln_lognormal_prime(dva * br2, rate); break;
            }
        }
        a_la[v] = term_bd.d[0];
        a_mu[v] = term_bd.d[1];
        a_va[v] = term_ck.d[1];
        gh[v] = term_bd.d[2] + term_ck.d[2];
        ep[v] = term_bd.d[3] + term_ck.d[3];
        g_R[b * lds + v] = term_ck.d[0];
    }
    // the value is the one of mcd_prior_logprior_batch (same code); the dual values above only carry the tangents.  (A loop of
    // its own: in one loop with the duals the two do not fit the registers of two waves per SIMD.)
    ClockCache cc;
    prior_clock_scalars(va, cc);
    for (int v = tid; v < n; v += nthr) {
        if (v == 0) continue;
        vb[v] = prior_bd_term(P, v, near, la, mu, h);
        vc[v] = prior_clock_term(P, v, va, cc.lg_k, cc.log_t, h, r);
    }
    __syncthreads();
    // gather the children's contributions (pre-order: first child = v + 1, the second follows the first one's sub tree)
    for (int v = tid; v < n; v += nthr) {
        double acc = gh[v];
        const int nc = P.n_children[v];
        if (nc > 0) {
            const int c1 = P.first_child[v];
            acc += ep[c1];
            if (nc > 1) acc += ep[P.second_child[v]];
        }
        gh[v] = acc;                                          // (only this thread reads or writes gh[v] here; ep is read-only)
    }
    __syncthreads();
    double total = 0.0, gla = 0.0, gmu = 0.0, gva = 0.0, s_th = 0.0;
    bool ok = false;
    if (wave == 0) {
    // ---- soft node priors: value lane-parallel (as prior_nodes_wave), derivative added by lane 0 in table order ----
    const double c0 = prior_nodes_wave(P, lane, th, h);
    if (lane == 0) {
        const double x = 1.0 / th;
        for (int i = 0; i < P.n_cal; ++i) {
            const int v = P.cal_node[i];
            const double hv = h[v];
            if (hv < 0) continue;
            if (P.cal_has_lo[i]) {
                const double a = (th == 1) ? P.cal_lo[i] : x * P.cal_lo[i];
                if (hv < a) {
                    const double s = 0.7978845608028654 * P.cal_lo_p[i];
                    const double w = (a - hv) / (s * s);              // d/d h_v of -1/2 ((a - h_v) / s)^2
                    gh[v] += w;
                    s_th += w * (P.cal_lo[i] / (th * th));            // a = lo / tH
                }
            }
            if (P.cal_has_hi[i]) {
                const double bb = (th == 1) ? P.cal_hi[i] : x * P.cal_hi[i];
                if (hv > bb) {
                    const double s = 0.7978845608028654 * P.cal_hi_p[i];
                    const double w = (hv - bb) / (s * s);
                    gh[v] -= w;
                    s_th -= w * (P.cal_hi[i] / (th * th));
                }
            }
        }
        for (int i = 0; i < P.n_con; ++i) {
            const int y = P.con_young[i], o = P.con_old[i];
            const double hy = h[y], ho = h[o];
            if (!(hy < ho)) {
                const double s = 0.7978845608028654 * P.con_p[i];
                const double w = (hy - ho) / (s * s);
                gh[y] -= w;
                gh[o] += w;
            }
        }
        for (int i = 0; i < P.n_brace; ++i) {
            const int lo = P.brace_ptr[i], hi = P.brace_ptr[i + 1];
            const double h0 = h[P.brace_nodes[lo]];
            bool all_equal = true;
            double sum = 0.0;
            for (int j = lo; j < hi; ++j) {
                const double hj = h[P.brace_nodes[j]];
                all_equal = all_equal && (hj == h0);
                sum += hj;
            }
            if (!all_equal) {
                const double mean = sum / (double)(hi - lo), sd = P.brace_sd[i];
                for (int j = lo; j < hi; ++j) gh[P.brace_nodes[j]] -= (h[P.brace_nodes[j]] - mean) / (sd * sd);
            }
        }
    }
    // the sums, lane by lane in the one-wave kernel's order: tangents over v = lane + 64 it (v = 0 carries none) ...
    double s_la = 0.0, s_mu = 0.0, s_va = 0.0;
    for (int v = lane; v < n; v += 64) {
        if (v == 0) continue;
        s_la += a_la[v];
        s_mu += a_mu[v];
        s_va += a_va[v];
    }
    gla = pr_wave_sum(s_la);
    gmu = pr_wave_sum(s_mu);
    gva = pr_wave_sum(s_va);
    // ... values over v = 1 + lane + 64 it, closed as prior_bd_wave / prior_clock_wave close them
    double bd = 0.0, clock = 0.0;
    for (int v = 1 + lane; v < n; v += 64) bd += vb[v];
    for (int v = 1 + lane; v < n; v += 64) clock += vc[v];
    total = c0 + prior_bd_finish(pr_wave_sum(bd), la, mu) + prior_clock_finish(P, pr_wave_sum(clock), rm, va, cc.hyper);
    ok = total == total && total > kNegInf;                   // inside the support
    if (lane == 0) bc[0] = ok ? 1.0 : 0.0;
    }
    __syncthreads();                                          // lane 0's additions to gh, the verdict
    ok = bc[0] != 0.0;
    const double bad = __builtin_nan("");
    for (int v = tid; v < n; v += nthr) {
        g_H[b * lds + v] = ok ? gh[v] : bad;
        if (!ok) g_R[b * lds + v] = bad;
    }
    if (tid == 0) {
        lp[b] = total;
        g_birth[b] = ok ? gla - 1.0 : bad;                    // d/d la [ln exponential 1 la] = -1
        g_death[b] = ok ? gmu - 1.0 : bad;
        g_tH[b] = ok ? s_th : bad;
        g_rMu[b] = ok ? -P.ht : bad;                          // d/d rMu [ln exponential ht rMu]
        g_rVar[b] = ok ? gva + (0.5 / va - 6.0) : bad;        // + d/d va [ln gamma(3/2, 1/6)(va)]
    }
}

hipError_t launch_prior_grad(const PriorDev& P, const double* birth, const double* death, const double* tH, const double* H,
                             const double* rMu, const double* rVar, const double* Rt, int64_t lds, int64_t batch, double* lp,
                             double* g_birth, double* g_death, double* g_tH, double* g_H, double* g_rMu, double* g_rVar, double* g_R,
                             hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    const size_t bytes = sizeof(double) * (7 * (size_t)P.n_nodes + 2);
    if (bytes > 64 * 1024) return hipErrorInvalidValue;
    // waves per chain: as many as the tree has 64-node slices (at most 4) while the whole batch is resident at once (2 waves
    // per SIMD: 2048 on the chip); beyond that more waves per chain only add rounds (4096 chains x 255 nodes: 115 us with one
    // wave per chain, 143 with four)
    int waves = (P.n_nodes + 63) / 64;
    const int64_t fit = 2048 / batch;
    waves = waves > 4 ? 4 : waves;
    waves = waves > fit ? (int)fit : waves;
    waves = waves < 1 ? 1 : waves;
    hipLaunchKernelGGL(k_prior_grad, dim3((unsigned)batch), dim3(64 * waves), bytes, st, P, birth, death, tH, H, rMu, rVar, Rt, lds, batch, lp,
                       g_birth, g_death, g_tH, g_H, g_rMu, g_rVar, g_R);
    return hipGetLastError();
}

}  // namespace mcd
