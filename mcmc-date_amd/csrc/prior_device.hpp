// prior_device.hpp -- wave-level evaluation of McmcDate's log prior (shared by k_prior.hip and k_mh_chain.hip).
//
// priorFunction ht md cb cs bs x  (app/Probability.hs:127-150); see k_prior.hip for the mapping and the citations of
// every factor.  `h` and `r` may point to global memory or LDS (generic pointers): node heights and branch rates of
// ONE chain.  All 64 lanes of the calling wave must be active; every lane returns the same value.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mvn_kernels.h"

namespace mcd {


__device__ __forceinline__ double pr_readlane64(double v, int l)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double pr_dpp64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pr_wave_sum(double v)
{
    v += pr_dpp64<0xB1>(v);
    v += pr_dpp64<0x4E>(v);
    v += pr_dpp64<0x124>(v);
    v += pr_dpp64<0x128>(v);
    return (pr_readlane64(v, 0) + pr_readlane64(v, 16)) + (pr_readlane64(v, 32) + pr_readlane64(v, 48));
}

constexpr double kLnSqrt2Pi = 0.9189385332046727417803297364056176;
constexpr double kNegInf = -__builtin_huge_val();

// [third party: mcmc, Mcmc.Prior] exponential / gamma densities, log domain
__device__ __forceinline__ double ln_exponential(double l, double x) { return (x < 0) ? kNegInf : log(l) + (-l * x); }
__device__ __forceinline__ double ln_gamma_pdf(double k, double t, double x)
{
    return (x <= 0) ? kNegInf : log(x) * (k - 1.0) - (x / t) - lgamma(k) - log(t) * k;
}
// d x / d 0 for d = normal 0 s  (Calibration.hs:391, Constraint.hs:414, Brace.hs:226-230)
__device__ __forceinline__ double ln_normal_ratio(double s, double x)
{
    const double q = x / s;
    return -0.5 * q * q;
}
// logNormal' m v x -- Prior/Branch/RelaxedClock.hs:141-150
__device__ __forceinline__ double ln_lognormal_prime(double m, double v, double x)
{
    if (x <= 0) return kNegInf;
    const double t = -(kLnSqrt2Pi + log(x * sqrt(v)));
    const double a = 1.0 / (2 * v);
    const double b = log(x / m) + 0.5 * v;
    return t + (-(a * b * b));
}
// computeDE / computeDENearCritical -- Prior/BirthDeath.hs:53-79, 90-114
__device__ __forceinline__ void compute_de(bool near, double la, double mu, double rho, double dt, double e0, double& pD,
                                           double& pE)
{
    const double d = la - mu;
    const double c = (1 - rho) + rho * e0;
    if (near) {
        const double y = (mu - c * la) * dt;
        const double denom = 1 + y;
        pD = (1 - d * dt) / denom / denom;
        pE = (c + y) / denom;
    } else {
        const double x = exp(-d * dt);
        const double y = (mu - c * la) * x;
        const double c1 = c - 1;
        const double denom = la * c1 + y;
        pD = d * d * x / denom / denom;
        pE = (mu * c1 + y) / denom;
    }
}

// The log prior is the sum of three blocks (what app/Monitor.hs monitors); each is a function of part of the state:
//   nodes  (th, heights)                      calibrations, constraints, braces           Combined.hs:70-92
//   bd     (la, mu, heights)                  birth-death prior + exponential 1 la, mu    app/Probability.hs:66-85
//   clock  (rm, va, rates; heights for the white-noise and autocorrelated models)         app/Probability.hs:96-124
// A caller that knows which part of the state changed re-evaluates only the affected blocks (k_mh_chain.hip).
__device__ __forceinline__ double prior_nodes_wave(const PriorDev& P, int lane, double th, const double* h)
{
    // ---- soft node priors ------------------------------------------------------------------------
    double node = 0.0;
    const double x = 1.0 / th;                                 // transformInterval (recip h), Calibration.hs:426-430
    for (int i = lane; i < P.n_cal; i += 64) {
        const double hv = h[P.cal_node[i]];
        double t = 0.0;
        if (hv < 0) {
            t = kNegInf;
        } else {
            if (P.cal_has_lo[i]) {
                const double a = (th == 1) ? P.cal_lo[i] : x * P.cal_lo[i];
                if (hv < a) t += ln_normal_ratio(0.7978845608028654 * P.cal_lo_p[i], a - hv);
            }
            if (P.cal_has_hi[i]) {
                const double bb = (th == 1) ? P.cal_hi[i] : x * P.cal_hi[i];
                if (hv > bb) t += ln_normal_ratio(0.7978845608028654 * P.cal_hi_p[i], hv - bb);
            }
        }
        node += t;
    }
    for (int i = lane; i < P.n_con; i += 64) {
        const double hy = h[P.con_young[i]], ho = h[P.con_old[i]];
        if (!(hy < ho)) node += ln_normal_ratio(0.7978845608028654 * P.con_p[i], hy - ho);
    }
    for (int i = lane; i < P.n_brace; i += 64) {
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
            const double mean = sum / (double)(hi - lo);
            for (int j = lo; j < hi; ++j) node += ln_normal_ratio(P.brace_sd[i], h[P.brace_nodes[j]] - mean);
        }
    }
    double c0 = pr_wave_sum(node);
    if (th <= 0) c0 = kNegInf;                                 // Combined.hs:78
    return c0;
}

// The calibration and constraint tables (a dozen entries in a typical analysis) in LDS: a persistent sampler kernel evaluates the node
// priors at every step that writes a height, and from global memory that is two dependent round trips per evaluation (8 calibrations +
// 4 constraints: +0.35 us per lock step at 257 and at 1025 nodes).  Doubles of LDS needed; 0 = nothing to stage.
// (prior_node_tables_doubles: mvn_kernels.h)
// every calling thread copies its share (tid of nthreads); Pl's pointers are redirected; the caller synchronises before the first use
__device__ __forceinline__ void prior_stage_node_tables(PriorDev& Pl, const PriorDev& P, double* lds, int tid, int nthreads)
{
    const int nc = P.n_cal, nk = P.n_con;
    double* c_lo = lds;
    double* c_lop = c_lo + nc;
    double* c_hi = c_lop + nc;
    double* c_hip = c_hi + nc;
    double* k_p = c_hip + nc;
    int32_t* c_node = reinterpret_cast<int32_t*>(k_p + nk);
    int32_t* c_hl = c_node + nc;
    int32_t* c_hh = c_hl + nc;
    int32_t* k_y = c_hh + nc;
    int32_t* k_o = k_y + nk;
    for (int i = tid; i < nc; i += nthreads) {
        c_lo[i] = P.cal_lo[i];
        c_lop[i] = P.cal_lo_p[i];
        c_hi[i] = P.cal_hi[i];
        c_hip[i] = P.cal_hi_p[i];
        c_node[i] = P.cal_node[i];
        c_hl[i] = P.cal_has_lo[i];
        c_hh[i] = P.cal_has_hi[i];
    }
    for (int i = tid; i < nk; i += nthreads) {
        k_p[i] = P.con_p[i];
        k_y[i] = P.con_young[i];
        k_o[i] = P.con_old[i];
    }
    Pl.cal_lo = c_lo;
    Pl.cal_lo_p = c_lop;
    Pl.cal_hi = c_hi;
    Pl.cal_hi_p = c_hip;
    Pl.con_p = k_p;
    Pl.cal_node = c_node;
    Pl.cal_has_lo = c_hl;
    Pl.cal_has_hi = c_hh;
    Pl.con_young = k_y;
    Pl.con_old = k_o;
}

// The birth-death and the clock blocks are sums over the nodes v = 1 .. n_nodes - 1, taken lane by lane over v = 1 + lane + 64 it
// (it ascending) and then over the wave.  The summand of one node and the closing scalar terms are functions of their own, so
// that a caller may deal the nodes to several waves and add the summands up in the same order (k_prior_grad.hip): the same
// function values, the same order, the same bits.
__device__ __forceinline__ double prior_bd_term(const PriorDev& P, int v, bool near, double la, double mu, const double* h)
{
    const int pv = P.parent[v];
    const double br = h[pv] - h[v];                            // heightTreeToLengthTree
    // E at the bottom of v's branch
    double e_bottom = 0.0;
    const int nc = P.n_children[v];
    if (nc > 0 && !near) {
        const double xx = exp(-(la - mu) * h[v]);
        e_bottom = mu * (1.0 - xx) / (la - mu * xx);
    } else if (nc > 0) {   // near-critical: compose branch by branch like the reference, tip first
        int u = P.first_child[v];
        int depth = 1;
        while (P.n_children[u] > 0) { u = P.first_child[u]; ++depth; }
        double e = 0.0;                                        // below a tip: E = 0 with the tip's sampling rate
        for (int i = 0; i < depth; ++i) {                      // u climbs from the tip to first_child[v]
            const double bu = h[P.parent[u]] - h[u];
            if (bu <= 0) {
                e = 1.0;                                       // `| br <= 0 = (0.0, 1.0)`
            } else {
                double dd, ee;
                compute_de(near, la, mu, 1.0, bu, e, dd, ee);   // rho = 1 everywhere in priorFunctionBirthDeath
                e = ee;
            }
            u = P.parent[u];
        }
        e_bottom = e;
    }
    if (br <= 0) return kNegInf;
    double dT, eT;
    compute_de(near, la, mu, 1.0, br, e_bottom, dT, eT);
    return log(dT * ((nc == 2) ? la : 1.0));                   // internal node: dT * la; tip / unary: dT * rho, rho = 1
}

__device__ __forceinline__ bool prior_bd_near(double la, double mu) { return 1e-6 > fabs(la - mu); }   // epsNearCritical, BirthDeath.hs:117-118

// c1 from the wave sum of the summands
__device__ __forceinline__ double prior_bd_finish(double c1, double la, double mu)
{
    if (la < 0 || mu < 0) c1 = __builtin_nan("");              // birthDeath: `error` on negative rates
    c1 += ln_exponential(1.0, la) + ln_exponential(1.0, mu);   // app/Probability.hs:72-73
    return c1;
}

__device__ __forceinline__ double prior_bd_wave(const PriorDev& P, int lane, double la, double mu, const double* h)
{
    const bool near = prior_bd_near(la, mu);
    double bd = 0.0;
    for (int v = 1 + lane; v < P.n_nodes; v += 64) bd += prior_bd_term(P, v, near, la, mu, h);
    return prior_bd_finish(pr_wave_sum(bd), la, mu);
}

// Wave-uniform pieces of the clock block that depend on the rate variance only: lgamma and two logarithms.  A caller
// that evaluates many states with the same variance (most proposals do not touch it) passes the cache of the previous
// evaluation; the values are the same function results either way.
struct ClockCache {
    double va, lg_k, log_t, hyper;    // lgamma(1 / va), log(va), ln gamma(3/2, 1/6)(va)
};

__device__ __forceinline__ void prior_clock_scalars(double va, ClockCache& c)
{
    c.va = va;
    c.lg_k = lgamma(1.0 / va);
    c.log_t = log(va);
    c.hyper = ln_gamma_pdf(1.5, 1.0 / 6.0, va);                            // app/Probability.hs:108-111
}

__device__ __forceinline__ double prior_clock_term(const PriorDev& P, int v, double va, double lg_k, double log_t, const double* h, const double* r)
{
    const double br = h[P.parent[v]] - h[v];                   // heightTreeToLengthTree
    // relaxed molecular clock, branchesWith WithoutStem (Prior/Branch.hs:23-25)
    const double rate = r[v];
    double term;
    switch (P.clock_model) {
        case 0: {                                              // uncorrelatedGamma 1.0 va = gamma (1 / va) va
            const double k = 1.0 / va;
            term = (rate <= 0) ? kNegInf : log(rate) * (k - 1.0) - (rate / va) - lg_k - log_t * k;
        } break;
        case 1: term = ln_lognormal_prime(1.0, va, rate); break;                      // uncorrelatedLogNormal
        case 2: { const double v2 = va / br; term = ln_gamma_pdf(1.0 / v2, v2, rate); } break;   // white noise
        default: term = ln_lognormal_prime(1.0, va * br, rate); break;                // autocorrelatedLogNormal
    }
    return term;
}

__device__ __forceinline__ double prior_clock_finish(const PriorDev& P, double c2, double rm, double va, double hyper)
{
    if (va <= 0) c2 = __builtin_nan("");                       // the reference calls `error` (variance <= 0)
    c2 += ln_exponential(P.ht, rm) + hyper;                    // :105-111
    return c2;
}

__device__ __forceinline__ double prior_clock_wave(const PriorDev& P, int lane, double rm, double va, const double* h,
                                                   const double* r, ClockCache* cache = nullptr)
{
    ClockCache c;
    if (cache != nullptr && cache->va == va) {
        c = *cache;
    } else {
        prior_clock_scalars(va, c);
        if (cache != nullptr) *cache = c;
    }
    double clock = 0.0;
    for (int v = 1 + lane; v < P.n_nodes; v += 64) clock += prior_clock_term(P, v, va, c.lg_k, c.log_t, h, r);
    return prior_clock_finish(P, pr_wave_sum(clock), rm, va, c.hyper);
}

// ln prior of one chain; c_out (may be null): node priors, birth-death block, clock block
__device__ __forceinline__ double prior_eval_wave(const PriorDev& P, int lane, double la, double mu, double th, double rm,
                                                  double va, const double* h, const double* r, double* c_out)
{
    const double c0 = prior_nodes_wave(P, lane, th, h);
    const double c1 = prior_bd_wave(P, lane, la, mu, h);
    const double c2 = prior_clock_wave(P, lane, rm, va, h, r);
    if (c_out) {
        c_out[0] = c0;
        c_out[1] = c1;
        c_out[2] = c2;
    }
    return c0 + c1 + c2;
}

}  // namespace mcd
