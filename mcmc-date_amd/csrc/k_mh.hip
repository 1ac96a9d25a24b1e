// k_mh.hip -- lock-step Metropolis-Hastings-Green steps for a batch of independent chains (gfx950).
// SURVEY.md 8(f) row f2.  All chains execute the same proposal of the cycle at the same time, each with its own
// counter-based random numbers, tuning parameter and accept/reject decision; the posterior of the proposed states
// is evaluated by the batched prior and likelihood kernels between `mh_propose` and `mh_accept`.
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

#include "../../include/mcmcdate_mvn.h"
#include "mvn_kernels.h"

namespace mcd {

namespace {

struct Rng {
    uint32_t k0, k1, chain, s0, s1;
};

__device__ __forceinline__ void philox_block(const Rng& g, uint32_t d, double& ua, double& ub)
{
    uint32_t c0 = d, c1 = g.chain, c2 = g.s0, c3 = g.s1, k0 = g.k0, k1 = g.k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        c0 = h1 ^ c1 ^ k0;
        c1 = l1;
        c2 = h0 ^ c3 ^ k1;
        c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    ua = ((double)((((uint64_t)c0 << 32) | c1) >> 11) + 0.5) * 0x1p-53;
    ub = ((double)((((uint64_t)c2 << 32) | c3) >> 11) + 0.5) * 0x1p-53;
}

__device__ __forceinline__ double phi2(double x) { return 0.5 * (1.0 + erf(x * 0.70710678118654752440)); }

// ln density of the normal(m, s) truncated to [a, b] at x; NaN where the reference raises `error`
__device__ double tn_logpdf(double m, double s, double a, double b, double x)
{
    if (!(s > 0) || !(a < b) || a > m || b < m) return __builtin_nan("");
    if (x < a || x > b) return -__builtin_huge_val();
    const double pa = phi2((a - m) / s), z = phi2((b - m) / s) - pa, xi = (x - m) / s;
    return log((1.0 / s) * (1.0 / z) * (0.39894228040143267794 * exp(-0.5 * xi * xi)));
}

// truncatedNormalSample: new value and ln (qYX / qXY)
__device__ void tn_sample(double m, double s, double t, double a, double b, double U, double& x, double& lnq)
{
    const double s1 = t * s;
    double u = __builtin_nan("");
    if ((s1 > 0) && (a < b) && !(a > m) && !(b < m)) {
        const double pa = phi2((a - m) / s1), z = phi2((b - m) / s1) - pa;
        u = erfinv(2.0 * (U * z + pa) - 1.0) * 1.41421356237309504880 * s1 + m;
    }
    if (!(a <= u && u <= b)) {
        x = __builtin_nan("");
        lnq = __builtin_nan("");
        return;
    }
    x = u;
    lnq = tn_logpdf(u, s1, a, b, m) - tn_logpdf(m, s1, a, b, u);
}

// Gamma(shape, scale), Marsaglia & Tsang
__device__ double gamma_sample(const Rng& g, double shape, double scale)
{
    double boost = 1.0, a = shape;
    if (a < 1.0) {
        double ua, ub;
        philox_block(g, 0xFFFFFFFEu, ua, ub);
        boost = pow(ua, 1.0 / a);
        a += 1.0;
    }
    const double dd = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * dd);
    for (uint32_t it = 0; it < 1000; ++it) {
        double u1a, u1b, u2a, u2b;
        philox_block(g, 2 * it, u1a, u1b);
        philox_block(g, 2 * it + 1, u2a, u2b);
        const double z = sqrt(-2.0 * log(u1a)) * cos(6.28318530717958647692 * u1b);
        const double v0 = 1.0 + c * z;
        if (v0 <= 0) continue;
        const double v = v0 * v0 * v0;
        if (log(u2a) < 0.5 * z * z + dd - dd * v + dd * log(v)) return dd * v * boost * scale;
    }
    return __builtin_nan("");
}

__device__ __forceinline__ double gamma_ratio(double k, double th, double u)
{
    return -2.0 * (k - 1.0) * log(u) - (1.0 / u - u) / th;
}

__device__ __forceinline__ double mh_readlane64(double v, int l)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double mh_wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace

// Writes the proposed state of every chain and ln(q-ratio * Jacobian) without the root-branch factor.
__global__ __launch_bounds__(256) void k_mh_propose(MhDev M, const int32_t* __restrict__ sched, int64_t sched_idx,
                                                    uint64_t step, uint64_t seed)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= M.batch) return;
    const int p = sched[sched_idx];
    const int n = M.n_nodes, kind = M.kind[p], v = M.node[p];
    const double p0 = M.p0[p], t = M.tune[b * M.n_prop + p];
    const int64_t B = M.batch;
    const double* H = M.H + b * M.ld;
    const double* R = M.R + b * M.ld;
    double* H1 = M.H1 + b * M.ld;
    double* R1 = M.R1 + b * M.ld;
    Rng g{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(M.chain0 + b), (uint32_t)step, (uint32_t)(step >> 32)};
    double sc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) sc[i] = M.sc[i * B + b];
    double lnq = 0.0, lnj = 0.0;
    // per-node transforms: H1[w] = (w in [hlo, hhi)) ? H[w] * hmul : H[w], with point overrides; R1[w] = R[w] * rmul + radd in [rlo, rhi)
    int hlo = 0, hhi = 0, hlo2 = 0, hhi2 = 0, rlo = 0, rhi = 0;
    double hmul = 1.0, hmul2 = 1.0, rmul = 1.0, radd = 0.0;
    int pt1 = -1, pt2 = -1;
    double pv1 = 0.0, pv2 = 0.0;
    bool rate_positive_guard = false;
    switch (kind) {
        case MCD_PROP_SCALE_SCALAR: {
            const double k = p0 / t, th = t / p0, u = gamma_sample(g, k, th);
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (i == v) sc[i] *= u;
            lnq = gamma_ratio(k, th, u);
            lnj = -log(u);
            break;
        }
        case MCD_PROP_SLIDE_NODE: {
            double hc = -__builtin_huge_val();
            const int end = v + M.size[v];
            for (int c = v + 1; c < end; c += M.size[c]) hc = fmax(hc, H[c]);
            const double hp = (v == 0) ? __builtin_huge_val() : H[M.parent[v]];
            double ua, ub, h1;
            philox_block(g, 0, ua, ub);
            tn_sample(H[v], p0, t, hc, hp, ua, h1, lnq);
            pt1 = v;
            pv1 = h1;
            break;
        }
        case MCD_PROP_SCALE_SUBTREE_TIME: {
            const double hp = (v == 0) ? __builtin_huge_val() : H[M.parent[v]];
            double ua, ub, h1;
            philox_block(g, 0, ua, ub);
            tn_sample(H[v], p0, t, 0.0, hp, ua, h1, lnq);
            const double xi = h1 / H[v];
            hlo = v + 1;
            hhi = v + M.size[v];
            hmul = xi;
            pt1 = v;
            pv1 = h1;
            lnj = (double)(M.n1[p] - 1) * log(xi);
            break;
        }
        case MCD_PROP_PULLEY: {
            const int l = 1, r = 1 + M.size[1];
            const double ht = H[0], hL = H[l], hR = H[r], brL = ht - hL, brR = ht - hR;
            if (!(brL > 0) || !(brR > 0)) {
                lnq = __builtin_nan("");
                break;
            }
            const double a = -fmin(brL, ht - brR), bb = fmin(brR, ht - brL);
            double ua, ub, u;
            philox_block(g, 0, ua, ub);
            tn_sample(0.0, p0, t, a, bb, ua, u, lnq);
            const double hL1 = hL - u, hR1 = hR + u, xiL = hL1 / hL, xiR = hR1 / hR;
            hlo = l + 1; hhi = l + M.size[l]; hmul = xiL; pt1 = l; pv1 = hL1;
            hlo2 = r + 1; hhi2 = r + M.size[r]; hmul2 = xiR; pt2 = r; pv2 = hR1;
            lnj = (double)(M.n1[p] - 1) * log(xiL) + (double)(M.n2[p] - 1) * log(xiR);
            break;
        }
        case MCD_PROP_SCALE_BRANCH_RATE: {
            const double k = p0 / t, th = t / p0, u = gamma_sample(g, k, th);
            rlo = v; rhi = v + 1; rmul = u;
            lnq = gamma_ratio(k, th, u);
            lnj = -log(u);
            break;
        }
        case MCD_PROP_SCALE_SUBTREE_RATE: {
            const double k = p0 / t, th = t / p0, u = gamma_sample(g, k, th);
            rlo = v; rhi = v + M.size[v]; rmul = u;
            lnq = gamma_ratio(k, th, u);
            lnj = (double)(M.n1[p] - 2) * log(u);
            break;
        }
        case MCD_PROP_SCALE_NORM_TREE: {
            const double k = p0 / t, th = t / p0, u = gamma_sample(g, k, th);
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (i == v) sc[i] /= u;
            rlo = 1; rhi = n; rmul = u;
            lnq = gamma_ratio(k, th, u);
            lnj = (double)((n - 1) - 2 - 1) * log(u);
            break;
        }
        case MCD_PROP_SCALE_VAR_TREE: {
            const double k = p0 / t, th = t / p0, u = gamma_sample(g, k, th);
            double s = 0.0;
            for (int w = lane; w < n; w += 64) s += (w >= 1) ? R[w] : 0.0;
            s = mh_wave_sum(s);
            const int nb = n - 1;
            const double mu = s / (double)nb, n1 = 1.0 / (double)nb;
            sc[4] = sc[4] * u * u;
            // (r - mu) u + mu: keep the reference's order of operations
            rlo = 1; rhi = n; rmul = u; radd = mu; rate_positive_guard = true;
            lnq = gamma_ratio(k, th, u);
            lnj = (double)nb * log(u - n1 * u + n1);
            break;
        }
        case MCD_PROP_SCALE_VAR_TREE_AUTO: {
            // The reference recursion y_v = y_parent + u (r_v - r_parent), anchored at rMu for the children of the
            // root, telescopes to y_v = rMu + u (r_v - rMu): one independent expression per lane (equal up to rounding).
            const double k = p0 / t, th = t / p0, u = gamma_sample(g, k, th);
            sc[4] = sc[4] * u * u;
            rlo = 1; rhi = n; rmul = u; radd = sc[3]; rate_positive_guard = true;
            lnq = gamma_ratio(k, th, u);
            lnj = (double)(n - 1) * log(u);
            break;
        }
        case MCD_PROP_SCALE_CONTRARILY: {
            const double k = p0 / t, th = M.p1[p] * t, u = gamma_sample(g, k, th);
            sc[2] *= u;
            sc[3] /= u;
            lnq = gamma_ratio(k, th, u);
            lnj = -2.0 * log(u);
            break;
        }
        default: lnq = __builtin_nan("");
    }
    for (int w = lane; w < n; w += 64) {
        double h = H[w], r = R[w];
        if (w >= hlo && w < hhi) h *= hmul;
        if (w >= hlo2 && w < hhi2) h *= hmul2;
        if (w == pt1) h = pv1;
        if (w == pt2) h = pv2;
        if (w >= rlo && w < rhi) {
            if (rate_positive_guard) {
                r = (r - radd) * rmul + radd;
                r = (r > 0) ? r : __builtin_nan("");
            } else {
                r *= rmul;
            }
        }
        H1[w] = h;
        R1[w] = r;
    }
    if (lane < 5) {
        double mine = sc[0];
#pragma unroll
        for (int i = 1; i < 5; ++i)
            if (lane == i) mine = sc[i];
        M.sc1[lane * B + b] = mine;
    }
    if (lane == 0) M.lnqj[b] = lnq + lnj;
}

// Accept or reject; post = (ln prior, ln likelihood, ln jacobianRootBranch), [3][batch].
__global__ __launch_bounds__(256) void k_mh_accept(MhDev M, const int32_t* __restrict__ sched, int64_t sched_idx,
                                                   uint64_t step, uint64_t seed, double* __restrict__ trace_alpha,
                                                   int8_t* __restrict__ trace_accept)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= M.batch) return;
    const int p = sched[sched_idx];
    const int64_t B = M.batch;
    const double lp = M.post[b], ll = M.post[B + b], lj = M.post[2 * B + b];
    const double lp1 = M.post1[b], ll1 = M.post1[B + b], lj1 = M.post1[2 * B + b];
    double la = (lp1 + ll1) - (lp + ll) + M.lnqj[b];
    if (M.jac_root[p]) la += lj1 - lj;
    Rng g{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(M.chain0 + b), (uint32_t)step, (uint32_t)(step >> 32)};
    double ua, ub;
    philox_block(g, 0xFFFFFFFFu, ua, ub);
    const bool ok = (la >= 0) || (ua < exp(la));
    if (ok) {
        const double* H1 = M.H1 + b * M.ld;
        const double* R1 = M.R1 + b * M.ld;
        double* H = M.H + b * M.ld;
        double* R = M.R + b * M.ld;
        for (int w = lane; w < M.n_nodes; w += 64) {
            H[w] = H1[w];
            R[w] = R1[w];
        }
        if (lane < 5) M.sc[lane * B + b] = M.sc1[lane * B + b];
        if (lane < 3) M.post[lane * B + b] = M.post1[lane * B + b];
    }
    if (lane == 0) {
        const int64_t i = b * M.n_prop + p;
        M.tried[i] += 1;
        if (ok) M.acc[i] += 1;
        if (trace_alpha) trace_alpha[b] = la;
        if (trace_accept) trace_accept[b] = ok ? 1 : 0;
    }
}

// After every iteration: running sums of the absolute node ages tH * h_v (the quantity the reference's
// node-age summaries report, scripts/trees-monitor-summary-ultrametric).
__global__ __launch_bounds__(256) void k_mh_accumulate(MhDev M)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M.batch * M.n_nodes) return;
    const int64_t b = i / M.n_nodes;
    const int v = (int)(i - b * M.n_nodes);
    const double a = M.sc[2 * M.batch + b] * M.H[b * M.ld + v];
    M.age_sum[b * M.n_nodes + v] += a;
    M.age_sq[b * M.n_nodes + v] += a * a;
}

// mcmc's auto tuning [external]: t' = clamp(t exp(2 (rate - optimal(dim))), 1e-5, 1e3); counters reset.
__global__ __launch_bounds__(256) void k_mh_tune(MhDev M)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M.batch * M.n_prop) return;
    const int n_tried = M.tried[i];
    if (n_tried > 0) {
        const int dim = M.dim[i % M.n_prop];
        const double opt = (dim == 1) ? 0.44 : (dim == 2) ? 0.352 : (dim == 3) ? 0.316 : (dim == 4) ? 0.279 : (dim == 5) ? 0.275 : 0.234;
        const double r = (double)M.acc[i] / (double)n_tried;
        double t = M.tune[i] * exp(2.0 * (r - opt));
        t = (t < 1e-5) ? 1e-5 : (t > 1e3) ? 1e3 : t;
        M.tune[i] = t;
    }
    M.acc[i] = 0;
    M.tried[i] = 0;
}

hipError_t launch_mh_propose(const MhDev& M, const int32_t* sched, int64_t sched_idx, uint64_t step, uint64_t seed, hipStream_t st)
{
    const int wpb = 4;
    hipLaunchKernelGGL(k_mh_propose, dim3((unsigned)((M.batch + wpb - 1) / wpb)), dim3(64 * wpb), 0, st, M, sched, sched_idx, step, seed);
    return hipGetLastError();
}
hipError_t launch_mh_accept(const MhDev& M, const int32_t* sched, int64_t sched_idx, uint64_t step, uint64_t seed,
                            double* trace_alpha, int8_t* trace_accept, hipStream_t st)
{
    const int wpb = 4;
    hipLaunchKernelGGL(k_mh_accept, dim3((unsigned)((M.batch + wpb - 1) / wpb)), dim3(64 * wpb), 0, st, M, sched, sched_idx, step, seed,
                       trace_alpha, trace_accept);
    return hipGetLastError();
}
hipError_t launch_mh_accumulate(const MhDev& M, hipStream_t st)
{
    const int64_t n = M.batch * M.n_nodes;
    hipLaunchKernelGGL(k_mh_accumulate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, M);
    return hipGetLastError();
}
hipError_t launch_mh_tune(const MhDev& M, hipStream_t st)
{
    const int64_t n = M.batch * M.n_prop;
    hipLaunchKernelGGL(k_mh_tune, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, M);
    return hipGetLastError();
}

}  // namespace mcd
