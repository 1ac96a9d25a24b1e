// mh_device.hpp -- wave-level pieces of the Metropolis-Hastings-Green step shared by k_mh.hip (one launch per step
// phase) and k_mh_chain.hip (whole schedule in one launch).  See k_mh.hip for the citations of every proposal.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mcmcdate_mvn.h"
#include "mvn_kernels.h"

namespace mcd {

// Hand-over words between two waves of a workgroup live in LDS and are polled: typed as LDS (address space 3), because a `volatile int*`
// made from a pointer into dynamic LDS compiles to FLAT loads with system scope, each followed by a wait for every outstanding global
// load of the wave (the next step's table row, say) as well.
typedef __attribute__((address_space(3))) volatile int lds_vint_t;
typedef __attribute__((address_space(3))) volatile double lds_vdouble_t;
__device__ __forceinline__ lds_vint_t* lds_vint(int* p) { return (lds_vint_t*)p; }
__device__ __forceinline__ lds_vdouble_t* lds_vdouble(double* p) { return (lds_vdouble_t*)p; }
// LDS keeps a wave's accesses in order and serves one CU: publishing needs this wave's earlier LDS writes issued (they are: program
// order) and the compiler kept from moving them -- not the full workgroup-scope fence, which also waits for every global load in flight
__device__ __forceinline__ void lds_publish_fence() { __builtin_amdgcn_s_waitcnt(0xc07f); __asm__ volatile("" ::: "memory"); }
__device__ __forceinline__ void lds_acquire_fence() { __asm__ volatile("" ::: "memory"); }

// Random stream of one (chain, step): Philox4x32-10, counter = (draw, chain, step_lo, step_hi), key = seed.
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
__device__ inline double tn_logpdf(double m, double s, double a, double b, double x)
{
    if (!(s > 0) || !(a < b) || a > m || b < m) return __builtin_nan("");
    if (x < a || x > b) return -__builtin_huge_val();
    const double pa = phi2((a - m) / s), z = phi2((b - m) / s) - pa, xi = (x - m) / s;
    return log((1.0 / s) * (1.0 / z) * (0.39894228040143267794 * exp(-0.5 * xi * xi)));
}

__device__ __forceinline__ double tn_readlane64(double v, int l)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}

// truncatedNormalSample: new value and ln (qYX / qXY).  All 64 lanes call it with the same arguments; the four normal
// distribution functions, the two exponentials and the two logarithms it needs come in pairs with different arguments,
// so even and odd lanes evaluate one member of each pair and the results are exchanged with v_readlane: the same
// function values as the straight-line form (tn_logpdf(u, s1, a, b, m) - tn_logpdf(m, s1, a, b, u)) at half the calls.
__device__ inline void tn_sample(double m, double s, double t, double a, double b, double U, double& x, double& lnq)
{
    const bool odd = (threadIdx.x & 1) != 0;
    const double s1 = t * s;
    double u = __builtin_nan("");
    double pa = 0.0, pb = 0.0;
    const bool valid = (s1 > 0) && (a < b) && !(a > m) && !(b < m);
    if (valid) {
        const double p1 = phi2(((odd ? b : a) - m) / s1);              // even lanes: Phi(alpha), odd lanes: Phi(beta)
        pa = tn_readlane64(p1, 0);
        pb = tn_readlane64(p1, 1);
        const double z = pb - pa;
        u = erfinv(2.0 * (U * z + pa) - 1.0) * 1.41421356237309504880 * s1 + m;
    }
    if (!(a <= u && u <= b)) {
        x = __builtin_nan("");
        lnq = __builtin_nan("");
        return;
    }
    x = u;
    const double p2 = phi2(((odd ? b : a) - u) / s1);                  // the same two for the reverse move (mean u)
    const double qa = tn_readlane64(p2, 0), qb = tn_readlane64(p2, 1);
    // even lanes: ln density of the forward move (mean m at u), odd lanes: of the reverse move (mean u at m)
    const double zz = odd ? (qb - qa) : (pb - pa);
    const double xi = odd ? (m - u) / s1 : (u - m) / s1;
    const double ld = log((1.0 / s1) * (1.0 / zz) * (0.39894228040143267794 * exp(-0.5 * xi * xi)));
    lnq = tn_readlane64(ld, 1) - tn_readlane64(ld, 0);
}

// Gamma(shape, scale), Marsaglia & Tsang
__device__ inline double gamma_sample(const Rng& g, double shape, double scale)
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


__device__ __forceinline__ Rng mh_rng(uint64_t seed, int64_t chain, uint64_t step)
{
    return Rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)chain, (uint32_t)step, (uint32_t)(step >> 32)};
}

// One row of the proposal table, by value (kernel argument or prefetched registers) so that a step does not start with
// a chain of dependent loads: schedule -> row -> node tables -> state.
struct PropRow {
    int kind, node, n1, n2, jac_root;
    double p0, p1;
};

__device__ __forceinline__ PropRow mh_load_row(const MhDev& M, int p)
{
    return PropRow{M.kind[p], M.node[p], M.n1[p], M.n2[p], M.jac_root[p], M.p0[p], M.p1[p]};
}

// Loads that travel a step AHEAD in a persistent kernel (the schedule's next entries, the next row of the proposal table, its tuning
// parameter) must be VECTOR loads: with a uniform address the compiler issues scalar loads, those return out of order and share their
// counter with LDS -- so the next wait for ANY LDS read (a few instructions later) waits for the prefetch as well, and every step pays a
// round trip to L2.  mh_vzero() is a zero the compiler cannot see through: an index plus it is a per-lane address, the load a vector
// load (its own counter, in order).  The loaded integers are made scalars again where they are consumed (mh_row_scalar).
__device__ __forceinline__ int mh_vzero()
{
    int z;
    __asm__ volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}
__device__ __forceinline__ PropRow mh_load_row_ahead(const MhDev& M, int p_plus_vzero)
{
    const int p = p_plus_vzero;
    return PropRow{M.kind[p], M.node[p], M.n1[p], M.n2[p], M.jac_root[p], M.p0[p], M.p1[p]};
}
__device__ __forceinline__ PropRow mh_row_scalar(const PropRow& r)
{
    return PropRow{__builtin_amdgcn_readfirstlane(r.kind), __builtin_amdgcn_readfirstlane(r.node), __builtin_amdgcn_readfirstlane(r.n1),
                   __builtin_amdgcn_readfirstlane(r.n2), __builtin_amdgcn_readfirstlane(r.jac_root), r.p0, r.p1};
}

// The STATE-INDEPENDENT random part of one step: what can be drawn knowing only the proposal row and its tuning
// parameter.  Gamma-multiplier proposals: the multiplier u, ln (q(1/u) / q(u)) and ln u; truncated-normal proposals:
// the uniform that goes through the quantile; every step: the acceptance uniform.  The whole-schedule kernel computes
// these for 64 consecutive steps at once, one step per lane (k_mh_chain.hip); k_mh_step per step.
struct StepDraws {
    double u, lnq, logu;   // gamma kinds
    double U;              // truncated-normal kinds: first double of block 0
    double Uacc;           // acceptance: first double of block 0xFFFFFFFF
};

__device__ __forceinline__ bool mh_is_gamma_kind(int kind)
{
    return kind == MCD_PROP_SCALE_SCALAR || kind == MCD_PROP_SCALE_BRANCH_RATE || kind == MCD_PROP_SCALE_SUBTREE_RATE ||
           kind == MCD_PROP_SCALE_NORM_TREE || kind == MCD_PROP_SCALE_VAR_TREE || kind == MCD_PROP_SCALE_VAR_TREE_AUTO ||
           kind == MCD_PROP_SCALE_CONTRARILY;
}

__device__ inline StepDraws mh_step_draws(const PropRow& row, double t, const Rng& g)
{
    StepDraws d{1.0, 0.0, 0.0, 0.5, 0.5};
    double ub;
    const int kind = row.kind;
    if (mh_is_gamma_kind(kind)) {
        const double p0 = row.p0;
        const double k = p0 / t, th = (kind == MCD_PROP_SCALE_CONTRARILY) ? row.p1 * t : t / p0;
        d.u = gamma_sample(g, k, th);
        d.lnq = gamma_ratio(k, th, d.u);
        d.logu = log(d.u);
    } else {
        philox_block(g, 0, d.U, ub);
    }
    philox_block(g, 0xFFFFFFFFu, d.Uacc, ub);
    return d;
}

// What a proposal does to the nodes, as the per-node transform mh_propose_node applies: the scalar part of a proposal (random draw,
// bounds, ratio, Jacobian) is one wave's work, the transform itself is independent per node -- a kernel with several waves per
// chain lets all of them apply it (k_mh_step_wg).
struct PropApply {
    int kind;
    int hlo, hhi, hlo2, hhi2, rlo, rhi;
    int pt1, pt2, rp1, rp2, rp3;
    int brace_lo, brace_hi;
    int rate_positive_guard, h_divide;
    double hmul, hmul2, rmul, radd, pv1, pv2, rm1, rm2, rm3, brace_delta;
};

// The integer fields of a transform are the same in every lane: as scalars (SGPRs) they turn the range tests and the rarely taken paths
// of mh_propose_node (the division, the guarded rates, the braces' loop) into scalar branches instead of predicated vector code.
__device__ __forceinline__ void mh_apply_scalars(PropApply& A)
{
    A.kind = __builtin_amdgcn_readfirstlane(A.kind);
    A.hlo = __builtin_amdgcn_readfirstlane(A.hlo);
    A.hhi = __builtin_amdgcn_readfirstlane(A.hhi);
    A.hlo2 = __builtin_amdgcn_readfirstlane(A.hlo2);
    A.hhi2 = __builtin_amdgcn_readfirstlane(A.hhi2);
    A.rlo = __builtin_amdgcn_readfirstlane(A.rlo);
    A.rhi = __builtin_amdgcn_readfirstlane(A.rhi);
    A.pt1 = __builtin_amdgcn_readfirstlane(A.pt1);
    A.pt2 = __builtin_amdgcn_readfirstlane(A.pt2);
    A.rp1 = __builtin_amdgcn_readfirstlane(A.rp1);
    A.rp2 = __builtin_amdgcn_readfirstlane(A.rp2);
    A.rp3 = __builtin_amdgcn_readfirstlane(A.rp3);
    A.brace_lo = __builtin_amdgcn_readfirstlane(A.brace_lo);
    A.brace_hi = __builtin_amdgcn_readfirstlane(A.brace_hi);
    A.rate_positive_guard = __builtin_amdgcn_readfirstlane(A.rate_positive_guard);
    A.h_divide = __builtin_amdgcn_readfirstlane(A.h_divide);
}

// proposed height and rate of node w
__device__ __forceinline__ void mh_propose_node(const MhDev& M, const PropApply& A, int w, const double* H, const double* R, double& h_out,
                                                double& r_out)
{
    double h = H[w], r = R[w];
    if (w >= A.hlo && w < A.hhi) h = A.h_divide ? h / A.hmul : h * A.hmul;
    if (w >= A.hlo2 && w < A.hhi2) h *= A.hmul2;
    if (w == A.pt1) h = A.pv1;
    if (w == A.pt2) h = A.pv2;
    if (w >= A.rlo && w < A.rhi) {
        if (A.rate_positive_guard) {
            r = (r - A.radd) * A.rmul + A.radd;
            r = (r > 0) ? r : __builtin_nan("");
        } else {
            r *= A.rmul;
        }
    }
    if (w == A.rp1) r *= A.rm1;
    if (w == A.rp2) r *= A.rm2;
    if (w == A.rp3) r *= A.rm3;
    for (int i = A.brace_lo; i < A.brace_hi; ++i) {          // braced nodes: w is one of them and / or a daughter of one
        const int x = M.brace_nodes[i];
        const double hN = H[x];
        if (w == x) {
            h = hN + A.brace_delta;
            if (A.kind == MCD_PROP_SLIDE_BRACE_CONTRA) {
                const double hP = H[M.parent[x]];
                r *= (hP - hN) / (hP - hN - A.brace_delta);
            }
        }
        if (A.kind == MCD_PROP_SLIDE_BRACE_CONTRA && M.parent[w] == x) r *= (hN - H[w]) / (hN + A.brace_delta - H[w]);
    }
    h_out = h;
    r_out = r;
}

// The scalar part of the proposal `row` with tuning parameter t on the state (sc, H, R) of one chain; all 64 lanes active.
// Updates sc in place, fills the per-node transform and returns ln (q-ratio * Jacobian) without the root-branch factor (NaN =
// invalid proposal => reject).
__device__ __forceinline__ double mh_propose_params(const MhDev& M, const PropRow& row, double t, const StepDraws& dr, int lane, double (&sc)[5],
                                                    const double* H, const double* R, PropApply& A)
{
    const int n = M.n_nodes, kind = row.kind, v = row.node;
    const double p0 = row.p0;
    double lnq = 0.0, lnj = 0.0;
    // per-node transforms: H1[w] = (w in [hlo, hhi)) ? H[w] * hmul : H[w], with point overrides; R1[w] = R[w] * rmul + radd in [rlo, rhi)
    int hlo = 0, hhi = 0, hlo2 = 0, hhi2 = 0, rlo = 0, rhi = 0;
    double hmul = 1.0, hmul2 = 1.0, rmul = 1.0, radd = 0.0;
    int pt1 = -1, pt2 = -1;
    double pv1 = 0.0, pv2 = 0.0;
    bool rate_positive_guard = false, h_divide = false;
    int rp1 = -1, rp2 = -1, rp3 = -1;          // single branches whose rate is multiplied (children, then the stem)
    double rm1 = 1.0, rm2 = 1.0, rm3 = 1.0;
    int brace_lo = 0, brace_hi = 0;
    double brace_delta = 0.0;
    switch (kind) {
        case MCD_PROP_SCALE_SCALAR: {
            const double u = dr.u;
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (i == v) sc[i] *= u;
            lnq = dr.lnq;
            lnj = -dr.logu;
            break;
        }
        case MCD_PROP_SLIDE_NODE: {
            double hc = -__builtin_huge_val();
            const int end = v + M.size[v];
            for (int c = v + 1; c < end; c += M.size[c]) hc = fmax(hc, H[c]);
            const double hp = (v == 0) ? __builtin_huge_val() : H[M.parent[v]];
            double ua, ub, h1;
            ua = dr.U;
            (void)ub;
            tn_sample(H[v], p0, t, hc, hp, ua, h1, lnq);
            pt1 = v;
            pv1 = h1;
            break;
        }
        case MCD_PROP_SCALE_SUBTREE_TIME: {
            const double hp = (v == 0) ? __builtin_huge_val() : H[M.parent[v]];
            double ua, ub, h1;
            ua = dr.U;
            (void)ub;
            tn_sample(H[v], p0, t, 0.0, hp, ua, h1, lnq);
            const double xi = h1 / H[v];
            hlo = v + 1;
            hhi = v + M.size[v];
            hmul = xi;
            pt1 = v;
            pv1 = h1;
            lnj = (double)(row.n1 - 1) * log(xi);
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
            ua = dr.U;
            (void)ub;
            tn_sample(0.0, p0, t, a, bb, ua, u, lnq);
            const double hL1 = hL - u, hR1 = hR + u, xiL = hL1 / hL, xiR = hR1 / hR;
            hlo = l + 1; hhi = l + M.size[l]; hmul = xiL; pt1 = l; pv1 = hL1;
            hlo2 = r + 1; hhi2 = r + M.size[r]; hmul2 = xiR; pt2 = r; pv2 = hR1;
            lnj = (double)(row.n1 - 1) * log(xiL) + (double)(row.n2 - 1) * log(xiR);
            break;
        }
        case MCD_PROP_SCALE_BRANCH_RATE: {
            const double u = dr.u;
            rlo = v; rhi = v + 1; rmul = u;
            lnq = dr.lnq;
            lnj = -dr.logu;
            break;
        }
        case MCD_PROP_SCALE_SUBTREE_RATE: {
            const double u = dr.u;
            rlo = v; rhi = v + M.size[v]; rmul = u;
            lnq = dr.lnq;
            lnj = (double)(row.n1 - 2) * dr.logu;
            break;
        }
        case MCD_PROP_SCALE_NORM_TREE: {
            const double u = dr.u;
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (i == v) sc[i] /= u;
            rlo = 1; rhi = n; rmul = u;
            lnq = dr.lnq;
            lnj = (double)((n - 1) - 2 - 1) * dr.logu;
            break;
        }
        case MCD_PROP_SCALE_VAR_TREE: {
            const double u = dr.u;
            double s = 0.0;
            for (int w = lane; w < n; w += 64) s += (w >= 1) ? R[w] : 0.0;
            s = mh_wave_sum(s);
            const int nb = n - 1;
            const double mu = s / (double)nb, n1 = 1.0 / (double)nb;
            sc[4] = sc[4] * u * u;
            // (r - mu) u + mu: keep the reference's order of operations
            rlo = 1; rhi = n; rmul = u; radd = mu; rate_positive_guard = true;
            lnq = dr.lnq;
            // reference: product of the diagonal of the Jacobian matrix, (u - u/n + 1/n)^n (Unconstrained.hs:321-326);
            // p1 = 1 selects the determinant u^(n-1) instead (the mean-preserving map has eigenvalues u (n-1 times) and 1)
            lnj = (row.p1 == 1.0) ? (double)(nb - 1) * dr.logu : (double)nb * log(u - n1 * u + n1);
            break;
        }
        case MCD_PROP_SCALE_VAR_TREE_AUTO: {
            // The reference recursion y_v = y_parent + u (r_v - r_parent), anchored at rMu for the children of the
            // root, telescopes to y_v = rMu + u (r_v - rMu): one independent expression per lane (equal up to rounding).
            const double u = dr.u;
            sc[4] = sc[4] * u * u;
            rlo = 1; rhi = n; rmul = u; radd = sc[3]; rate_positive_guard = true;
            lnq = dr.lnq;
            lnj = (double)(n - 1) * dr.logu;
            break;
        }
        case MCD_PROP_SCALE_CONTRARILY: {
            const double u = dr.u;
            sc[2] *= u;
            sc[3] /= u;
            lnq = dr.lnq;
            lnj = -2.0 * dr.logu;
            break;
        }
        case MCD_PROP_SLIDE_NODE_CONTRA: {   // Contrary.hs:35-77
            double hc = -__builtin_huge_val();
            const int end = v + M.size[v];
            for (int c = v + 1; c < end; c += M.size[c]) hc = fmax(hc, H[c]);
            const double hN = H[v], hP = H[M.parent[v]];
            double ua, ub, h1;
            ua = dr.U;
            (void)ub;
            tn_sample(hN, p0, t, hc, hP, ua, h1, lnq);
            pt1 = v;
            pv1 = h1;
            const double xiStem = (hP - hN) / (hP - h1);
            double sumlog = 0.0;
            int k = 0;
            for (int c = v + 1; c < end; c += M.size[c], ++k) {
                const double xi = (hN - H[c]) / (h1 - H[c]);
                sumlog += log(xi);
                if (k == 0) { rp1 = c; rm1 = xi; } else { rp2 = c; rm2 = xi; }
            }
            rp3 = v;
            rm3 = xiStem;
            lnj = sumlog + log(xiStem);
            break;
        }
        case MCD_PROP_SCALE_SUBTREE_CONTRA: {   // Contrary.hs:269-326
            const double hN = H[v], hP = H[M.parent[v]];
            double ua, ub, h1;
            ua = dr.U;
            (void)ub;
            tn_sample(hN, p0, t, 0.0, hP, ua, h1, lnq);
            const double xiT = h1 / hN, xiR = 1.0 / xiT, xiStem = (hP - hN) / (hP - h1);
            hlo = v + 1; hhi = v + M.size[v]; hmul = xiT; pt1 = v; pv1 = h1;
            rlo = v + 1; rhi = v + M.size[v]; rmul = xiR;
            rp3 = v;
            rm3 = xiStem;
            lnj = (double)(row.n1 - row.n2) * log(xiT) + log(xiStem);
            break;
        }
        case MCD_PROP_SLIDE_ROOT_CONTRA: {   // Contrary.hs:191-223
            if (fabs(H[0] - 1.0) > 1e-14) {
                lnq = __builtin_nan("");
                break;
            }
            const int l = 1, r = 1 + M.size[1];
            const double ht = sc[2], hL = H[l], hR = H[r];
            double ua, ub, ht1;
            ua = dr.U;
            (void)ub;
            tn_sample(ht, p0, t, ht * fmax(hL, hR), __builtin_huge_val(), ua, ht1, lnq);
            const double u = ht1 / ht;
            const double xil = (1.0 - hL) / (u - hL), xir = (1.0 - hR) / (u - hR);
            hlo = 1; hhi = n; hmul = u; h_divide = true;
            rp1 = l; rm1 = xil; rp2 = r; rm2 = xir;
            sc[2] = ht1;
            lnj = (double)(-row.n1) * log(u) + log(xil) + log(xir);
            break;
        }
        case MCD_PROP_SCALE_RATES_TREE_CONTRA: {   // Contrary.hs:420-446 on (timeBirthRate, rateMean, timeTree)
            const int l = 1, r = 1 + M.size[1];
            const double m = fmax(H[l], H[r]);
            double ua, ub, m1;
            ua = dr.U;
            (void)ub;
            tn_sample(m, p0, t, 0.0, H[0], ua, m1, lnq);
            const double xi = m1 / m;
            hlo = 1; hhi = n; hmul = xi;
            sc[0] = sc[0] / xi;
            sc[3] = sc[3] / xi;
            lnj = (double)(row.n1 - 1 - 2) * log(xi);
            break;
        }
        case MCD_PROP_SLIDE_BRACE:
        case MCD_PROP_SLIDE_BRACE_CONTRA: {   // Brace.hs:98-156, 37-61; the per-node updates are part of the copy loop below
            const int lo = M.brace_ptr[v], hi = M.brace_ptr[v + 1];
            double a = -__builtin_huge_val(), bb = __builtin_huge_val();
            for (int i = lo; i < hi; ++i) {
                const int x = M.brace_nodes[i];
                double hc = -__builtin_huge_val();
                const int end = x + M.size[x];
                for (int c = x + 1; c < end; c += M.size[c]) hc = fmax(hc, H[c]);
                a = fmax(a, hc - H[x]);
                bb = fmin(bb, H[M.parent[x]] - H[x]);
            }
            double ua, ub;
            ua = dr.U;
            (void)ub;
            tn_sample(0.0, p0, t, a, bb, ua, brace_delta, lnq);
            brace_lo = lo;
            brace_hi = hi;
            if (kind == MCD_PROP_SLIDE_BRACE_CONTRA) {
                double sumlog = 0.0;
                for (int i = lo; i < hi; ++i) {
                    const int x = M.brace_nodes[i];
                    const double hN = H[x], hP = H[M.parent[x]];
                    sumlog += log((hP - hN) / (hP - hN - brace_delta));
                    const int end = x + M.size[x];
                    for (int c = x + 1; c < end; c += M.size[c]) sumlog += log((hN - H[c]) / (hN + brace_delta - H[c]));
                }
                lnj = sumlog;
            }
            break;
        }
        default: lnq = __builtin_nan("");
    }
    A = PropApply{kind, hlo, hhi, hlo2, hhi2, rlo, rhi, pt1, pt2, rp1, rp2, rp3, brace_lo, brace_hi, rate_positive_guard ? 1 : 0, h_divide ? 1 : 0,
                  hmul, hmul2, rmul, radd, pv1, pv2, rm1, rm2, rm3, brace_delta};
    return lnq + lnj;
}

// WHICH nodes the proposal `row` writes -- the integer fields of the PropApply mh_propose_params fills -- from the row and the topology
// alone: they do not depend on the state or on the draws (two kinds can bail out on an invalid state and then write nothing: whoever
// uses this ahead of the proposal compares it with the transform the proposal posts).  A wave that evaluates the likelihood beside the
// chain wave can so list the moved distances, and fetch what it needs for them, WHILE the proposal is being drawn.
__device__ __forceinline__ void mh_propose_ranges(const MhDev& M, int kind, int v, PropApply& A)
{
    const int n = M.n_nodes;
    int hlo = 0, hhi = 0, hlo2 = 0, hhi2 = 0, rlo = 0, rhi = 0, pt1 = -1, pt2 = -1, rp1 = -1, rp2 = -1, rp3 = -1, brace_lo = 0, brace_hi = 0;
    switch (kind) {
        case MCD_PROP_SLIDE_NODE: pt1 = v; break;
        case MCD_PROP_SCALE_SUBTREE_TIME: hlo = v + 1; hhi = v + M.size[v]; pt1 = v; break;
        case MCD_PROP_PULLEY: {
            const int l = 1, r = 1 + M.size[1];
            hlo = l + 1; hhi = l + M.size[l]; pt1 = l;
            hlo2 = r + 1; hhi2 = r + M.size[r]; pt2 = r;
            break;
        }
        case MCD_PROP_SCALE_BRANCH_RATE: rlo = v; rhi = v + 1; break;
        case MCD_PROP_SCALE_SUBTREE_RATE: rlo = v; rhi = v + M.size[v]; break;
        case MCD_PROP_SCALE_NORM_TREE:
        case MCD_PROP_SCALE_VAR_TREE:
        case MCD_PROP_SCALE_VAR_TREE_AUTO: rlo = 1; rhi = n; break;
        case MCD_PROP_SLIDE_NODE_CONTRA: {
            pt1 = v;
            const int end = v + M.size[v];
            int k = 0;
            for (int c = v + 1; c < end; c += M.size[c], ++k) {
                if (k == 0) rp1 = c; else rp2 = c;
            }
            rp3 = v;
            break;
        }
        case MCD_PROP_SCALE_SUBTREE_CONTRA: hlo = v + 1; hhi = v + M.size[v]; pt1 = v; rlo = v + 1; rhi = v + M.size[v]; rp3 = v; break;
        case MCD_PROP_SLIDE_ROOT_CONTRA: hlo = 1; hhi = n; rp1 = 1; rp2 = 1 + M.size[1]; break;
        case MCD_PROP_SCALE_RATES_TREE_CONTRA: hlo = 1; hhi = n; break;
        case MCD_PROP_SLIDE_BRACE:
        case MCD_PROP_SLIDE_BRACE_CONTRA: brace_lo = M.brace_ptr[v]; brace_hi = M.brace_ptr[v + 1]; break;
        default: break;                                      // the scalars, scaleContrarily: no node is written
    }
    A.kind = kind;
    A.hlo = hlo; A.hhi = hhi; A.hlo2 = hlo2; A.hhi2 = hhi2; A.rlo = rlo; A.rhi = rhi;
    A.pt1 = pt1; A.pt2 = pt2; A.rp1 = rp1; A.rp2 = rp2; A.rp3 = rp3;
    A.brace_lo = brace_lo; A.brace_hi = brace_hi;
}
__device__ __forceinline__ bool mh_same_ranges(const PropApply& a, const PropApply& b)
{
    return a.kind == b.kind && a.hlo == b.hlo && a.hhi == b.hhi && a.hlo2 == b.hlo2 && a.hhi2 == b.hhi2 && a.rlo == b.rlo && a.rhi == b.rhi && a.pt1 == b.pt1 &&
           a.pt2 == b.pt2 && a.rp1 == b.rp1 && a.rp2 == b.rp2 && a.rp3 == b.rp3 && a.brace_lo == b.brace_lo && a.brace_hi == b.brace_hi;
}
// does the proposal move tH * rMu (and with it every branch distance)?
__device__ __forceinline__ bool mh_moves_scale(int kind, int v)
{
    return (kind == MCD_PROP_SCALE_SCALAR && (v == 2 || v == 3)) || kind == MCD_PROP_SCALE_NORM_TREE || kind == MCD_PROP_SCALE_CONTRARILY ||
           kind == MCD_PROP_SLIDE_ROOT_CONTRA || kind == MCD_PROP_SCALE_RATES_TREE_CONTRA;
}

// Apply the proposal `row` with tuning parameter t to the state (sc, H, R) of one chain; all 64 lanes active.
// Writes the proposed heights / rates to H1 / R1 (global memory or LDS), updates sc in place and returns
// ln (q-ratio * Jacobian) without the root-branch factor (NaN = invalid proposal => reject).
__device__ __forceinline__ double mh_propose_wave(const MhDev& M, const PropRow& row, double t, const StepDraws& dr, int lane, double (&sc)[5],
                                                  const double* H, const double* R, double* H1, double* R1)
{
    PropApply A;
    const double lnqj = mh_propose_params(M, row, t, dr, lane, sc, H, R, A);
    for (int w = lane; w < M.n_nodes; w += 64) {
        double h, r;
        mh_propose_node(M, A, w, H, R, h, r);
        H1[w] = h;
        R1[w] = r;
    }
    return lnqj;
}

__device__ __forceinline__ double mh_optimal_rate(int dim)
{
    return (dim == 1) ? 0.44 : (dim == 2) ? 0.352 : (dim == 3) ? 0.316 : (dim == 4) ? 0.279 : (dim == 5) ? 0.275 : 0.234;
}

}  // namespace mcd
