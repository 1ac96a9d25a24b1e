/*
 * mh_oracle.c -- CPU twin of the batched Metropolis-Hastings-Green driver.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as mvn_oracle.c): the product never links or calls this file.
 *
 * What it restates, one chain at a time and strictly sequentially:
 *   - the proposals of lib/Mcmc/Tree/Proposal/Ultrametric.hs (slide node :50-59, scale sub tree :126-149,
 *     pulley :221-286) and Unconstrained.hs (scaleTree :95-106, scaleNormAndTreeContrarily :221-256,
 *     scaleVarianceAndTree :286-316, scaleVarianceAndTreeAutocorrelated :354-386), Contrary.hs (slide nodes
 *     contrarily, slide root contrarily, scale sub tree contrarily, scale rates and tree contrarily) and Brace.hs
 *     (slide braced nodes, ultrametric and contrarily);
 *   - truncatedNormalSample (lib/Mcmc/Tree/Proposal/Internal.hs:107-138) over
 *     lib/Statistics/Distribution/TruncatedNormal.hs:55-130 (density, quantile);
 *   - the generic gamma-multiplier proposals `scaleUnbiased k` and `scaleContrarily k th` and the MHG acceptance
 *     step of the `mcmc` package [EXTERNAL: dschrempf/mcmc rev 542c43f6, pinned by flake.lock; not vendored in
 *     the reference; restated from its documented behaviour: u ~ Gamma(k/t, th t), x' = x u, ratio
 *     q(1/u)/q(u), Jacobian 1/u (1/u^2 for the contrary pair); accept iff U < posterior ratio * q * J];
 *   - liftProposalWith jacobianRootBranch (app/Definitions.hs:148 ff.): J *= jf(y') / jf(y);
 *   - the auto-tuning rule of `mcmc` [EXTERNAL]: t' = clamp(t * exp(2 (rate - optimal(dim))), 1e-5, 1e3).
 *
 * PINNED AT THE SAMPLER LEVEL (round 3) by outputs the reference itself commits: the node ages of its six posterior and six
 * prior-only chains on the 7-taxon mtCDNApri analysis (bench/comparison_with_mcmctree/03_compare_estimates/*_samples_run*.tsv).
 * tests/test_reference_samples.py runs THIS twin (with mvn_oracle.c and prior_oracle.c underneath) on the same inputs, no GPU
 * involved: all six ages within 1 % of the reference's pooled means in both analyses -- the prior-only one with the root bound
 * the samples carry (30, not the committed file's 100: see that test) --, quantiles, standard deviations, correlations.
 * Single steps remain unpinned (the reference holds no expected outputs for one proposal and cannot be run here); what
 * pins them: Philox4x32-10 known answer (Random123 kat vector), scipy.stats.truncnorm / scipy.stats.gamma for the
 * distributions, detailed-balance identities (tests/test_mh_oracle.py).
 *
 * Random numbers are COUNTER BASED so that the device driver reproduces them exactly:
 *   block(seed, chain, step, d) = Philox4x32-10(counter = (d, chain, step_lo, step_hi), key = (seed_lo, seed_hi));
 *   the two doubles of a block are ((x0 << 32 | x1) >> 11 + 0.5) 2^-53 and the same from (x2, x3);
 *   proposal draws use d = 0, 1, 2, ...; the acceptance uniform is the first double of d = 0xFFFFFFFF.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* from mvn_oracle.c / prior_oracle.c */
int orc_tree_loglik_full(int n_nodes, const int32_t *parent, const double *heights, const double *rates, double tH,
                         double rMu, const double *mu, const double *sigma_inv, double logdet_sigma, double *ll);
int orc_log_jacobian_root_branch(int n_nodes, const int32_t *parent, const double *heights, const double *rates,
                                 double tH, double rMu, double *logj);
double orp_prior(double ht, int model, int n_nodes, const int32_t *parent, double birth, double death, double tH,
                 const double *heights, double rMu, double rVar, const double *rates, int ncal, const int32_t *cal_node,
                 const int32_t *cal_has_lo, const double *cal_lo, const double *cal_lo_p, const int32_t *cal_has_hi,
                 const double *cal_hi, const double *cal_hi_p, int ncon, const int32_t *con_young, const int32_t *con_old,
                 const double *con_p, int nbr, const int32_t *br_ptr, const int32_t *br_nodes, const double *br_sd,
                 double *components);

/* proposal kinds (shared numbering with include/mcmcdate_mvn.h MCD_PROP_*) */
enum {
    ORM_SCALE_SCALAR = 0,        /* node = which scalar (0 birth, 1 death, 2 tH, 3 rMu, 4 rVar); p0 = shape k        */
    ORM_SLIDE_NODE = 1,          /* node; p0 = sd                                                                   */
    ORM_SCALE_SUBTREE_TIME = 2,  /* node; p0 = sd; n1 = inner nodes of the sub tree                                 */
    ORM_PULLEY = 3,              /* p0 = sd; n1, n2 = inner nodes of the left / right sub tree                      */
    ORM_SCALE_BRANCH_RATE = 4,   /* node; p0 = shape                                                                */
    ORM_SCALE_SUBTREE_RATE = 5,  /* node; p0 = shape; n1 = nodes of the sub tree                                    */
    ORM_SCALE_NORM_TREE = 6,     /* node = which scalar (2 tH or 3 rMu); p0 = shape                                 */
    ORM_SCALE_VAR_TREE = 7,      /* p0 = shape                                                                      */
    ORM_SCALE_VAR_TREE_AUTO = 8, /* p0 = shape                                                                      */
    ORM_SCALE_CONTRARILY = 9,    /* (tH, rMu); p0 = shape k, p1 = scale th                                          */
    ORM_SLIDE_NODE_CONTRA = 10,  /* node; p0 = sd                                   Contrary.hs:35-77               */
    ORM_SCALE_SUBTREE_CONTRA = 11, /* node; p0 = sd; n1 = inner nodes, n2 = nodes of the sub tree   :269-326           */
    ORM_SLIDE_ROOT_CONTRA = 12,  /* p0 = sd; n1 = inner nodes of the tree           :191-223                        */
    ORM_SCALE_RATES_TREE_CONTRA = 13, /* p0 = sd; n1 = inner nodes - 1 (birth rate, rate mean, time tree)  :420-446    */
    ORM_SLIDE_BRACE = 14,        /* node = brace index; p0 = sd                     Brace.hs:98-156 (ultrametric)   */
    ORM_SLIDE_BRACE_CONTRA = 15  /* node = brace index; p0 = sd                     Brace.hs:37-61 ... contrarily   */
};

typedef struct {
    int32_t n_nodes;
    const int32_t *parent;
    /* likelihood operands */
    const double *mu, *sigma_inv;
    double logdet;
    /* prior */
    double ht;
    int32_t clock_model;
    int32_t ncal;
    const int32_t *cal_node, *cal_has_lo;
    const double *cal_lo, *cal_lo_p;
    const int32_t *cal_has_hi;
    const double *cal_hi, *cal_hi_p;
    int32_t ncon;
    const int32_t *con_young, *con_old;
    const double *con_p;
    int32_t nbr;
    const int32_t *br_ptr, *br_nodes;
    const double *br_sd;
    /* proposals */
    int32_t n_prop;
    const int32_t *kind, *node, *n1, *n2, *jac_root, *dim;
    const double *p0, *p1;
} orm_model;

/* ---- Philox4x32-10 (Salmon et al. 2011, Random123) ------------------------------------------------------- */
void orm_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1,
                       n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct { uint64_t seed, step; uint32_t chain, d; } orm_rng;

static void rng_block(orm_rng *g, uint32_t d, double u[2])
{
    const uint32_t ctr[4] = {d, g->chain, (uint32_t)g->step, (uint32_t)(g->step >> 32)};
    const uint32_t key[2] = {(uint32_t)g->seed, (uint32_t)(g->seed >> 32)};
    uint32_t x[4];
    orm_philox4x32(ctr, key, x);
    u[0] = ((double)((((uint64_t)x[0] << 32) | x[1]) >> 11) + 0.5) * 0x1p-53;
    u[1] = ((double)((((uint64_t)x[2] << 32) | x[3]) >> 11) + 0.5) * 0x1p-53;
}

void orm_uniform_pair(uint64_t seed, uint32_t chain, uint64_t step, uint32_t d, double *u)
{
    orm_rng g = {seed, step, chain, 0};
    rng_block(&g, d, u);
}

/* ---- special functions ------------------------------------------------------------------------------------- */
/* inverse error function: rational start (Giles 2010, single precision form) + Halley steps on erf/erfc to convergence */
double orm_erfinv(double y)
{
    if (!(y > -1.0 && y < 1.0)) return (y == 1.0) ? INFINITY : (y == -1.0) ? -INFINITY : NAN;
    if (y == 0.0) return 0.0;
    double w = -log((1.0 - y) * (1.0 + y)), x;
    if (w < 5.0) {
        w -= 2.5;
        x = 2.81022636e-08; x = 3.43273939e-07 + x * w; x = -3.5233877e-06 + x * w; x = -4.39150654e-06 + x * w;
        x = 0.00021858087 + x * w; x = -0.00125372503 + x * w; x = -0.00417768164 + x * w; x = 0.246640727 + x * w;
        x = 1.50140941 + x * w;
    } else {
        w = sqrt(w) - 3.0;
        x = -0.000200214257; x = 0.000100950558 + x * w; x = 0.00134934322 + x * w; x = -0.00367342844 + x * w;
        x = 0.00573950773 + x * w; x = -0.0076224613 + x * w; x = 0.00943887047 + x * w; x = 1.00167406 + x * w;
        x = 2.83297682 + x * w;
    }
    x *= y;
    const double two_over_sqrtpi = 1.1283791670955125738961589031215452;
    for (int it = 0; it < 12; ++it) {
        /* residual without cancellation: erf(x) - y for small |y|, (1 - y) - erfc(x) in the tails */
        const double f = (fabs(y) < 0.5) ? erf(x) - y : (y > 0 ? (1.0 - y) - erfc(x) : erfc(-x) - (1.0 + y));
        const double fp = two_over_sqrtpi * exp(-x * x);
        const double dx = f / fp;
        const double step = dx / (1.0 + x * dx);   /* Halley: f'' / f' = -2x */
        x -= step;
        if (fabs(step) <= 1e-17 * fabs(x)) break;
    }
    return x;
}

static double phi2(double x) { return 0.5 * (1.0 + erf(x * 0.70710678118654752440)); }   /* TruncatedNormal.hs:84-85 */

/* log density of truncatedNormalDistr m s a b at x -- TruncatedNormal.hs:55-79, 97-105 */
double orm_tn_logpdf(double m, double s, double a, double b, double x)
{
    if (!(s > 0) || !(a < b) || a > m || b < m) return NAN;   /* the reference calls `error` */
    if (x < a || x > b) return -INFINITY;
    const double pa = phi2((a - m) / s), z = phi2((b - m) / s) - pa, xi = (x - m) / s;
    return log((1.0 / s) * (1.0 / z) * (0.39894228040143267794 * exp(-0.5 * xi * xi)));
}

/* quantile -- TruncatedNormal.hs:117-130 */
double orm_tn_quantile(double m, double s, double a, double b, double p)
{
    if (!(s > 0) || !(a < b) || a > m || b < m) return NAN;
    if (p == 0) return a;
    if (p == 1) return b;
    const double pa = phi2((a - m) / s), z = phi2((b - m) / s) - pa;
    const double val = 2.0 * (p * z + pa) - 1.0;
    return orm_erfinv(val) * 1.41421356237309504880 * s + m;
}

/* truncatedNormalSample -- Internal.hs:107-138: value and ln (qYX / qXY); NaN when the reference would `error`. */
static void tn_sample(double m, double s, double t, double a, double b, double U, double *x, double *lnq)
{
    const double s1 = t * s;
    const double u = orm_tn_quantile(m, s1, a, b, U);
    if (!(a <= u && u <= b)) { *x = NAN; *lnq = NAN; return; }     /* "out of bounds" is an `error` upstream */
    const double qxy = orm_tn_logpdf(m, s1, a, b, u), qyx = orm_tn_logpdf(u, s1, a, b, m);
    *x = u;
    *lnq = qyx - qxy;
}

/* Gamma(shape, scale) by Marsaglia & Tsang (2000); shape < 1 through the U^(1/shape) boost.  Consumes blocks d0.. */
static double gamma_sample(orm_rng *g, double shape, double scale)
{
    double boost = 1.0, a = shape;
    if (a < 1.0) {
        double ub[2];
        rng_block(g, 0xFFFFFFFEu, ub);
        boost = pow(ub[0], 1.0 / a);
        a += 1.0;
    }
    const double dd = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * dd);
    for (uint32_t it = 0; it < 1000; ++it) {
        double u1[2], u2[2];
        rng_block(g, 2 * it, u1);
        rng_block(g, 2 * it + 1, u2);
        const double z = sqrt(-2.0 * log(u1[0])) * cos(6.28318530717958647692 * u1[1]);
        const double v0 = 1.0 + c * z;
        if (v0 <= 0) continue;
        const double v = v0 * v0 * v0;
        if (log(u2[0]) < 0.5 * z * z + dd - dd * v + dd * log(v)) return dd * v * boost * scale;
    }
    return NAN;
}

double orm_gamma_draw(uint64_t seed, uint32_t chain, uint64_t step, double shape, double scale)
{
    orm_rng g = {seed, step, chain, 0};
    return gamma_sample(&g, shape, scale);
}

/* ln [ gamma(k, th)(1/u) / gamma(k, th)(u) ] */
static double gamma_ratio(double k, double th, double u) { return -2.0 * (k - 1.0) * log(u) - (1.0 / u - u) / th; }

/* ---- one chain's state ------------------------------------------------------------------------------------- */
typedef struct {
    double sc[5];      /* birth, death, tH, rMu, rVar */
    double *H, *R;     /* [n_nodes] relative heights; rates (index 0 = stem, unused) */
} orm_state;

static void subtree_sizes(int n, const int32_t *parent, int32_t *size)
{
    for (int v = 0; v < n; ++v) size[v] = 1;
    for (int v = n - 1; v > 0; --v) size[parent[v]] += size[v];
}

/*
 * Apply proposal p to `cur`, writing `y`; returns ln q-ratio and ln Jacobian (NaN = invalid => reject).
 */
static void propose(const orm_model *M, const int32_t *size, int p, double t, const orm_state *cur, orm_state *y,
                    orm_rng *g, double *lnq, double *lnj)
{
    const int n = M->n_nodes;
    memcpy(y->sc, cur->sc, sizeof y->sc);
    memcpy(y->H, cur->H, sizeof(double) * (size_t)n);
    memcpy(y->R, cur->R, sizeof(double) * (size_t)n);
    const int v = M->node[p];
    const double p0 = M->p0[p];
    *lnq = 0.0;
    *lnj = 0.0;
    switch (M->kind[p]) {
        case ORM_SCALE_SCALAR: {   /* scaleUnbiased k: Gamma(k / t, t / k) */
            const double k = p0 / t, th = t / p0, u = gamma_sample(g, k, th);
            y->sc[v] = cur->sc[v] * u;
            *lnq = gamma_ratio(k, th, u);
            *lnj = -log(u);
            break;
        }
        case ORM_SLIDE_NODE: {   /* Ultrametric.hs:50-59 */
            double hc = -INFINITY, ub[2];
            for (int w = v + 1; w < v + size[v]; ++w)
                if (M->parent[w] == v && cur->H[w] > hc) hc = cur->H[w];
            const double hp = (v == 0) ? INFINITY : cur->H[M->parent[v]];
            rng_block(g, 0, ub);
            double h1;
            tn_sample(cur->H[v], p0, t, hc, hp, ub[0], &h1, lnq);
            y->H[v] = h1;
            break;
        }
        case ORM_SCALE_SUBTREE_TIME: {   /* Ultrametric.hs:126-149 */
            double ub[2], h1;
            const double hp = (v == 0) ? INFINITY : cur->H[M->parent[v]];
            rng_block(g, 0, ub);
            tn_sample(cur->H[v], p0, t, 0.0, hp, ub[0], &h1, lnq);
            const double xi = h1 / cur->H[v];
            y->H[v] = h1;
            for (int w = v + 1; w < v + size[v]; ++w) y->H[w] = cur->H[w] * xi;
            *lnj = (double)(M->n1[p] - 1) * log(xi);
            break;
        }
        case ORM_PULLEY: {   /* Ultrametric.hs:221-286 */
            const int l = 1, r = 1 + size[1];
            const double ht = cur->H[0], hL = cur->H[l], hR = cur->H[r], brL = ht - hL, brR = ht - hR;
            if (!(brL > 0) || !(brR > 0)) { *lnq = NAN; break; }
            const double a = -fmin(brL, ht - brR), b = fmin(brR, ht - brL);
            double ub[2], u;
            rng_block(g, 0, ub);
            tn_sample(0.0, p0, t, a, b, ub[0], &u, lnq);
            const double hL1 = hL - u, hR1 = hR + u, xiL = hL1 / hL, xiR = hR1 / hR;
            y->H[l] = hL1;
            for (int w = l + 1; w < l + size[l]; ++w) y->H[w] = cur->H[w] * xiL;
            y->H[r] = hR1;
            for (int w = r + 1; w < r + size[r]; ++w) y->H[w] = cur->H[w] * xiR;
            *lnj = (double)(M->n1[p] - 1) * log(xiL) + (double)(M->n2[p] - 1) * log(xiR);
            break;
        }
        case ORM_SCALE_BRANCH_RATE: {
            const double k = p0 / t, th = t / p0, u = gamma_sample(g, k, th);
            y->R[v] = cur->R[v] * u;
            *lnq = gamma_ratio(k, th, u);
            *lnj = -log(u);
            break;
        }
        case ORM_SCALE_SUBTREE_RATE: {   /* Unconstrained.hs:84-106: every branch of the sub tree, stem included */
            const double k = p0 / t, th = t / p0, u = gamma_sample(g, k, th);
            for (int w = v; w < v + size[v]; ++w) y->R[w] = cur->R[w] * u;
            *lnq = gamma_ratio(k, th, u);
            *lnj = (double)(M->n1[p] - 2) * log(u);
            break;
        }
        case ORM_SCALE_NORM_TREE: {   /* Unconstrained.hs:221-256 */
            const double k = p0 / t, th = t / p0, u = gamma_sample(g, k, th);
            y->sc[v] = cur->sc[v] / u;
            for (int w = 1; w < n; ++w) y->R[w] = cur->R[w] * u;
            *lnq = gamma_ratio(k, th, u);
            *lnj = (double)((n - 1) - 2 - 1) * log(u);
            break;
        }
        case ORM_SCALE_VAR_TREE: {   /* Unconstrained.hs:286-316 */
            const double k = p0 / t, th = t / p0, u = gamma_sample(g, k, th);
            const int nb = n - 1;
            double s = 0;
            for (int w = 1; w < n; ++w) s += cur->R[w];
            const double mu = s / (double)nb, n1 = 1.0 / (double)nb;
            y->sc[4] = cur->sc[4] * u * u;
            for (int w = 1; w < n; ++w) {
                const double b1 = (cur->R[w] - mu) * u + mu;
                y->R[w] = (b1 > 0) ? b1 : NAN;
            }
            *lnq = gamma_ratio(k, th, u);
            *lnj = (M->p1[p] == 1.0) ? (double)(nb - 1) * log(u) : (double)nb * log(u - n1 * u + n1);   /* p1 = 1: the determinant */
            break;
        }
        case ORM_SCALE_VAR_TREE_AUTO: {   /* Unconstrained.hs:354-386: y_v = y_parent + u (r_v - r_parent), root level at rMu */
            const double k = p0 / t, th = t / p0, u = gamma_sample(g, k, th);
            const int nb = n - 1;
            double *yy = (double *)malloc(sizeof(double) * (size_t)n);
            for (int w = 1; w < n; ++w) {
                const int pa = M->parent[w];
                const double mu_old = (pa == 0) ? cur->sc[3] : cur->R[pa], mu_new = (pa == 0) ? cur->sc[3] : yy[pa];
                yy[w] = mu_new + u * (cur->R[w] - mu_old);
                y->R[w] = (yy[w] > 0) ? yy[w] : NAN;
            }
            free(yy);
            y->sc[4] = cur->sc[4] * u * u;
            *lnq = gamma_ratio(k, th, u);
            *lnj = (double)nb * log(u);
            break;
        }
        case ORM_SCALE_CONTRARILY: {   /* mcmc scaleContrarily k th on (tH, rMu) */
            const double k = p0 / t, th = M->p1[p] * t, u = gamma_sample(g, k, th);
            y->sc[2] = cur->sc[2] * u;
            y->sc[3] = cur->sc[3] / u;
            *lnq = gamma_ratio(k, th, u);
            *lnj = -2.0 * log(u);
            break;
        }
        case ORM_SLIDE_NODE_CONTRA: {   /* Contrary.hs:35-77 (the root is never handled: `hn` excludes the empty path) */
            double hc = -INFINITY, ub[2], h1;
            for (int w = v + 1; w < v + size[v]; ++w)
                if (M->parent[w] == v && cur->H[w] > hc) hc = cur->H[w];
            const double hN = cur->H[v], hP = cur->H[M->parent[v]];
            rng_block(g, 0, ub);
            tn_sample(hN, p0, t, hc, hP, ub[0], &h1, lnq);
            y->H[v] = h1;
            const double xiStem = (hP - hN) / (hP - h1);
            double sumlog = 0.0;
            for (int w = v + 1; w < v + size[v]; ++w)
                if (M->parent[w] == v) {
                    const double xi = (hN - cur->H[w]) / (h1 - cur->H[w]);
                    y->R[w] = cur->R[w] * xi;
                    sumlog += log(xi);
                }
            y->R[v] = cur->R[v] * xiStem;
            *lnj = sumlog + log(xiStem);
            break;
        }
        case ORM_SCALE_SUBTREE_CONTRA: {   /* Contrary.hs:269-326 */
            double ub[2], h1;
            const double hN = cur->H[v], hP = cur->H[M->parent[v]];
            rng_block(g, 0, ub);
            tn_sample(hN, p0, t, 0.0, hP, ub[0], &h1, lnq);
            const double xiT = h1 / hN, xiR = 1.0 / xiT, xiStem = (hP - hN) / (hP - h1);
            y->H[v] = h1;
            for (int w = v + 1; w < v + size[v]; ++w) {
                y->H[w] = cur->H[w] * xiT;
                y->R[w] = cur->R[w] * xiR;
            }
            y->R[v] = cur->R[v] * xiStem;
            *lnj = (double)(M->n1[p] - M->n2[p]) * log(xiT) + log(xiStem);
            break;
        }
        case ORM_SLIDE_ROOT_CONTRA: {   /* Contrary.hs:191-223 */
            if (fabs(cur->H[0] - 1.0) > 1e-14) { *lnq = NAN; break; }        /* `error` upstream */
            const int l = 1, r = 1 + size[1];
            const double ht = cur->sc[2], hmax = fmax(cur->H[l], cur->H[r]);
            double ub[2], ht1;
            rng_block(g, 0, ub);
            tn_sample(ht, p0, t, ht * hmax, INFINITY, ub[0], &ht1, lnq);
            const double u = ht1 / ht;
            const double xil = (1.0 - cur->H[l]) / (u - cur->H[l]), xir = (1.0 - cur->H[r]) / (u - cur->H[r]);
            for (int w = 1; w < n; ++w) y->H[w] = cur->H[w] / u;
            y->R[l] = cur->R[l] * xil;
            y->R[r] = cur->R[r] * xir;
            y->sc[2] = ht1;
            *lnj = (double)(-M->n1[p]) * log(u) + log(xil) + log(xir);
            break;
        }
        case ORM_SCALE_RATES_TREE_CONTRA: {   /* Contrary.hs:420-446 on (timeBirthRate, rateMean, timeTree), app/Definitions.hs:238-239 */
            const int l = 1, r = 1 + size[1];
            const double m = fmax(cur->H[l], cur->H[r]);
            double ub[2], m1;
            rng_block(g, 0, ub);
            tn_sample(m, p0, t, 0.0, cur->H[0], ub[0], &m1, lnq);
            const double xi = m1 / m;
            for (int w = 1; w < n; ++w) y->H[w] = cur->H[w] * xi;
            y->sc[0] = cur->sc[0] / xi;
            y->sc[3] = cur->sc[3] / xi;
            *lnj = (double)(M->n1[p] - 1 - 2) * log(xi);
            break;
        }
        case ORM_SLIDE_BRACE:
        case ORM_SLIDE_BRACE_CONTRA: {   /* Brace.hs:98-156 and :37-61 */
            const int lo = M->br_ptr[v], hi = M->br_ptr[v + 1];
            double a = -INFINITY, b = INFINITY, ub[2], delta;
            for (int i = lo; i < hi; ++i) {
                const int x = M->br_nodes[i];
                double hc = -INFINITY;
                for (int w = x + 1; w < x + size[x]; ++w)
                    if (M->parent[w] == x && cur->H[w] > hc) hc = cur->H[w];
                a = fmax(a, hc - cur->H[x]);
                b = fmin(b, cur->H[M->parent[x]] - cur->H[x]);
            }
            rng_block(g, 0, ub);
            tn_sample(0.0, p0, t, a, b, ub[0], &delta, lnq);
            double sumlog = 0.0;
            for (int i = lo; i < hi; ++i) {
                const int x = M->br_nodes[i];
                y->H[x] = cur->H[x] + delta;
                if (M->kind[p] == ORM_SLIDE_BRACE_CONTRA) {
                    const double hN = cur->H[x], hP = cur->H[M->parent[x]];
                    const double xiS = (hP - hN) / (hP - hN - delta);
                    y->R[x] = y->R[x] * xiS;
                    sumlog += log(xiS);
                    for (int w = x + 1; w < x + size[x]; ++w)
                        if (M->parent[w] == x) {
                            const double xi = (hN - cur->H[w]) / (hN + delta - cur->H[w]);
                            y->R[w] = y->R[w] * xi;
                            sumlog += log(xi);
                        }
                }
            }
            *lnj = sumlog;
            break;
        }
        default: *lnq = NAN;
    }
}

static void posterior(const orm_model *M, const orm_state *x, double *lp, double *ll, double *lj)
{
    *lp = orp_prior(M->ht, M->clock_model, M->n_nodes, M->parent, x->sc[0], x->sc[1], x->sc[2], x->H, x->sc[3], x->sc[4],
                    x->R, M->ncal, M->cal_node, M->cal_has_lo, M->cal_lo, M->cal_lo_p, M->cal_has_hi, M->cal_hi,
                    M->cal_hi_p, M->ncon, M->con_young, M->con_old, M->con_p, M->nbr, M->br_ptr, M->br_nodes, M->br_sd,
                    NULL);
    if (orc_tree_loglik_full(M->n_nodes, M->parent, x->H, x->R, x->sc[2], x->sc[3], M->mu, M->sigma_inv, M->logdet, ll)) *ll = NAN;
    if (orc_log_jacobian_root_branch(M->n_nodes, M->parent, x->H, x->R, x->sc[2], x->sc[3], lj)) *lj = NAN;
}

/*
 * Run n_iter iterations of S steps on `batch` chains (chain-major state arrays, in/out).
 *   sched[n_iter * S]      proposal id per step (all chains step together)
 *   tune, acc, tried       [batch][n_prop], in/out
 *   post[batch][3]         out: ln prior, ln likelihood, ln jacobianRootBranch of the final state
 *   age_sum, age_sq        [batch][n_nodes] or NULL: += tH * H[v] (and its square) after every iteration
 *   trace_alpha            [n_iter * S][batch] or NULL: ln acceptance ratio of every step
 *   trace_accept           [n_iter * S][batch] or NULL
 *   beta                   [batch] reciprocal temperatures of MC3's heated chains, or NULL (all 1)
 */
int orm_run(const orm_model *M, int64_t batch, double *birth, double *death, double *tH, double *H, double *rMu,
            double *rVar, double *R, int64_t ld, const int32_t *sched, int64_t n_iter, int32_t S, uint64_t seed,
            uint64_t step0, int64_t chain0, const double *beta, double *tune, int32_t *acc, int32_t *tried, double *post, double *age_sum, double *age_sq,
            double *trace_alpha, int8_t *trace_accept)
{
    const int n = M->n_nodes, P = M->n_prop;
    int32_t *size = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    subtree_sizes(n, M->parent, size);
    for (int64_t i = 0; i < n_iter * S; ++i)
        if (sched[i] < 0 || sched[i] >= P) { free(size); return -1; }
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < batch; ++b) {
        orm_state x, y;
        x.H = (double *)malloc(sizeof(double) * (size_t)n * 4);
        x.R = x.H + n; y.H = x.R + n; y.R = y.H + n;
        x.sc[0] = birth[b]; x.sc[1] = death[b]; x.sc[2] = tH[b]; x.sc[3] = rMu[b]; x.sc[4] = rVar[b];
        memcpy(x.H, H + b * ld, sizeof(double) * (size_t)n);
        memcpy(x.R, R + b * ld, sizeof(double) * (size_t)n);
        double lp, ll, lj;
        posterior(M, &x, &lp, &ll, &lj);
        for (int64_t it = 0; it < n_iter; ++it) {
            for (int s = 0; s < S; ++s) {
                const int64_t gs = it * S + s;
                const int p = sched[gs];
                orm_rng g = {seed, step0 + (uint64_t)gs, (uint32_t)(chain0 + b), 0};
                double lnq, lnj, lp1, ll1, lj1, ua[2];
                propose(M, size, p, tune[b * P + p], &x, &y, &g, &lnq, &lnj);
                posterior(M, &y, &lp1, &ll1, &lj1);
                double la = (beta ? beta[b] : 1.0) * ((lp1 + ll1) - (lp + ll)) + lnq + lnj;   /* MC3: posterior^beta */
                if (M->jac_root[p]) la += (double)M->jac_root[p] * (lj1 - lj);
                rng_block(&g, 0xFFFFFFFFu, ua);
                const int ok = (la >= 0) || (ua[0] < exp(la));   /* NaN compares false => reject */
                if (trace_alpha) trace_alpha[gs * batch + b] = la;
                if (trace_accept) trace_accept[gs * batch + b] = (int8_t)ok;
                tried[b * P + p] += 1;
                if (ok) {
                    acc[b * P + p] += 1;
                    memcpy(x.sc, y.sc, sizeof x.sc);
                    memcpy(x.H, y.H, sizeof(double) * (size_t)n);
                    memcpy(x.R, y.R, sizeof(double) * (size_t)n);
                    lp = lp1; ll = ll1; lj = lj1;
                }
            }
            if (age_sum)
                for (int v = 0; v < n; ++v) {
                    const double a = x.sc[2] * x.H[v];
                    age_sum[b * n + v] += a;
                    if (age_sq) age_sq[b * n + v] += a * a;
                }
        }
        birth[b] = x.sc[0]; death[b] = x.sc[1]; tH[b] = x.sc[2]; rMu[b] = x.sc[3]; rVar[b] = x.sc[4];
        memcpy(H + b * ld, x.H, sizeof(double) * (size_t)n);
        memcpy(R + b * ld, x.R, sizeof(double) * (size_t)n);
        if (post) { post[b * 3] = lp; post[b * 3 + 1] = ll; post[b * 3 + 2] = lj; }
        free(x.H);
    }
    free(size);
    return 0;
}

/* optimal acceptance rate by proposal dimension -- mcmc [EXTERNAL] getOptimalRate */
static double optimal_rate(int dim)
{
    switch (dim) {
        case 1: return 0.44;
        case 2: return 0.352;
        case 3: return 0.316;
        case 4: return 0.279;
        case 5: return 0.275;
        default: return 0.234;
    }
}

/* auto tuning after a tuning period: t' = clamp(t exp(2 (rate - optimal))), counters reset */
void orm_tune(const orm_model *M, int64_t batch, double *tune, int32_t *acc, int32_t *tried)
{
    const int P = M->n_prop;
    for (int64_t i = 0; i < batch * P; ++i) {
        if (tried[i] > 0) {
            const double r = (double)acc[i] / (double)tried[i];
            double t = tune[i] * exp(2.0 * (r - optimal_rate(M->dim[i % P])));
            if (t < 1e-5) t = 1e-5;
            if (t > 1e3) t = 1e3;
            tune[i] = t;
        }
        acc[i] = 0;
        tried[i] = 0;
    }
}

/* expose one proposal application for the identity tests (tests/test_mh_oracle.py) */
int orm_propose_once(const orm_model *M, int p, double t, uint64_t seed, uint32_t chain, uint64_t step, const double *sc,
                     const double *H, const double *R, double *sc1, double *H1, double *R1, double *lnq, double *lnj)
{
    const int n = M->n_nodes;
    int32_t *size = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    subtree_sizes(n, M->parent, size);
    orm_state x, y;
    memcpy(x.sc, sc, sizeof x.sc);
    x.H = (double *)H; x.R = (double *)R;
    y.H = H1; y.R = R1;
    orm_rng g = {seed, step, chain, 0};
    propose(M, size, p, t, &x, &y, &g, lnq, lnj);
    memcpy(sc1, y.sc, sizeof y.sc);
    free(size);
    return 0;
}
