/*
 * prior_oracle.c -- CPU restatement of McmcDate's prior function.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as mvn_oracle.c).
 *
 * Follows  app/Probability.hs:46-150  (priorFunction and its three factors),
 *          lib/Mcmc/Tree/Prior/Node/Combined.hs:70-92, Calibration.hs:369-391, 426-430,
 *          Constraint.hs:403-415, Brace.hs:218-230            (soft node priors),
 *          lib/Mcmc/Tree/Prior/BirthDeath.hs:53-239          (Stadler 2011 birth-death prior),
 *          lib/Mcmc/Tree/Prior/Branch.hs:23-25, Branch/RelaxedClock.hs:110-324  (relaxed clocks).
 *
 * PINNED for the birth-death part: the reference carries known-answer values in comments
 * (BirthDeath.hs:51-52, 252-271, partly cross-checked against RevBayes); tests/test_prior_oracle.py
 * checks this file against every one of them.  UNPINNED for the rest: the elementary densities
 * `exponential`, `gamma`, `normal`, `gammaMeanVarianceToShapeScale` come from the third-party `mcmc`
 * package (Mcmc.Prior, rev 542c43f, not vendored) and are restated from their documented definitions.
 *
 * All functions return LOG-domain values (the argument of `Exp`); probability 0 is -INFINITY.
 * Trees are pre-order parent arrays (see mvn_oracle.c).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define LN_SQRT_2_PI 0.9189385332046727417803297364056176

/* ---- [ext] mcmc: Mcmc.Prior ------------------------------------------------------------------ */
/* exponential l x | x < 0 = 0 | otherwise = Exp (log l) * Exp (negate l * x) */
double orp_ln_exponential(double l, double x) { return (x < 0) ? -INFINITY : log(l) + (-l * x); }
/* gamma k t x | x <= 0 = 0 | otherwise = Exp $ log x * (k - 1) - (x / t) - logGamma k - log t * k */
double orp_ln_gamma(double k, double t, double x)
{
    return (x <= 0) ? -INFINITY : log(x) * (k - 1.0) - (x / t) - lgamma(k) - log(t) * k;
}
/* normal m s x = Exp $ (-0.5) * ((x - m) / s)^2 - log s - ln sqrt (2 pi) */
double orp_ln_normal(double m, double s, double x)
{
    const double dx = (x - m) / s;
    return -0.5 * dx * dx - log(s) - LN_SQRT_2_PI;
}
/* gammaMeanVarianceToShapeScale m v = (m * m / v, v / m) */
static void shape_scale(double m, double v, double *k, double *th) { *k = m * m / v; *th = v / m; }

/* ---- soft node priors ------------------------------------------------------------------------- */
/* calibrateSoftF (Interval a' b') h -- Calibration.hs:369-391.  has_lo / has_hi: boundary present. */
double orp_calibrate_soft(int has_lo, double a, double pa, int has_hi, double b, double pb, double h)
{
    if (h < 0) return -INFINITY;                                   /* :371 */
    double lower = 0.0, upper = 0.0;
    if (has_lo && h < a) {                                         /* :376-379 */
        const double s = 0.7978845608028654 * pa;                  /* :391  sqrt (2/pi) * probability mass */
        lower = orp_ln_normal(0, s, a - h) - orp_ln_normal(0, s, 0);
    }
    if (has_hi && h > b) {                                         /* :382-385 */
        const double s = 0.7978845608028654 * pb;
        upper = orp_ln_normal(0, s, h - b) - orp_ln_normal(0, s, 0);
    }
    return lower + upper;
}
/* constrainSoftF p (hY, hO) -- Constraint.hs:403-415 */
double orp_constrain_soft(double p, double hY, double hO)
{
    if (hY < hO) return 0.0;
    const double s = 0.7978845608028654 * p;
    return orp_ln_normal(0, s, hY - hO) - orp_ln_normal(0, s, 0);
}
/* braceSoftF s hs -- Brace.hs:218-230 */
double orp_brace_soft(double s, int n, const double *hs)
{
    int all_equal = 1;
    for (int i = 1; i < n; ++i) if (hs[i] != hs[0]) all_equal = 0;
    if (all_equal) return 0.0;
    double sum = 0.0;
    for (int i = 0; i < n; ++i) sum += hs[i];
    const double mean = sum / (double)n;
    const double d0 = orp_ln_normal(0, s, 0);
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += orp_ln_normal(0, s, hs[i] - mean) - d0;
    return r;
}

/* calibrateConstrainBraceSoft h cs ks bs t -- Combined.hs:70-92.  The calibration intervals are
 * given in absolute time and transformed with 1/h (transformCalibration, Calibration.hs:426-430). */
double orp_node_priors(double h, const double *heights,
                       int ncal, const int32_t *cal_node, const int32_t *cal_has_lo, const double *cal_lo,
                       const double *cal_lo_p, const int32_t *cal_has_hi, const double *cal_hi, const double *cal_hi_p,
                       int ncon, const int32_t *con_young, const int32_t *con_old, const double *con_p,
                       int nbr, const int32_t *br_ptr, const int32_t *br_nodes, const double *br_sd)
{
    if (h <= 0) return -INFINITY;                                  /* :78 */
    double r = 0.0;
    const double x = 1.0 / h;                                      /* transformInterval (recip h) */
    for (int i = 0; i < ncal; ++i) {
        const double a = (h == 1) ? cal_lo[i] : x * cal_lo[i];     /* :429 `h == 1 = c` */
        const double b = (h == 1) ? cal_hi[i] : x * cal_hi[i];
        r += orp_calibrate_soft(cal_has_lo[i], a, cal_lo_p[i], cal_has_hi[i], b, cal_hi_p[i], heights[cal_node[i]]);
    }
    for (int i = 0; i < ncon; ++i) r += orp_constrain_soft(con_p[i], heights[con_young[i]], heights[con_old[i]]);
    for (int i = 0; i < nbr; ++i) {
        double hs[64];
        const int n = br_ptr[i + 1] - br_ptr[i];
        for (int j = 0; j < n && j < 64; ++j) hs[j] = heights[br_nodes[br_ptr[i] + j]];
        r += orp_brace_soft(br_sd[i], n, hs);
    }
    return r;
}

/* ---- birth-death prior ------------------------------------------------------------------------ */
/* computeDE la mu rho dt e0 -- BirthDeath.hs:53-79 */
void orp_compute_de(double la, double mu, double rho, double dt, double e0, double *pD, double *pE)
{
    const double d = la - mu;
    const double x = exp(-d * dt);
    const double c = (1 - rho) + rho * e0;
    const double y = (mu - c * la) * x;
    const double nomD = d * d * x;
    const double c1 = c - 1;
    const double nomE = mu * c1 + y;
    const double denom = la * c1 + y;
    *pD = nomD / denom / denom;
    *pE = nomE / denom;
}
/* computeDENearCritical -- BirthDeath.hs:90-114 */
void orp_compute_de_near_critical(double la, double mu, double rho, double dt, double e0, double *pD, double *pE)
{
    const double d = la - mu;
    const double c = (1 - rho) + rho * e0;
    const double y = (mu - c * la) * dt;
    const double nomD = 1 - d * dt;
    const double nomE = c + y;
    const double denom = 1 + y;
    *pD = nomD / denom / denom;
    *pE = nomE / denom;
}

typedef struct { double lnD, E; } DE;

/* birthDeathWith f la mu rho (subtree at v) -- BirthDeath.hs:186-239 */
static DE bd_with(int near, double la, double mu, double rho, int n_nodes, const int32_t *parent, const double *len, int v)
{
    DE out;
    const double br = len[v];
    if (br <= 0) { out.lnD = -INFINITY; out.E = 1.0; return out; }        /* `| br <= 0 = (0.0, 1.0)` */
    int ch[3], nc = 0;
    for (int c = v + 1; c < n_nodes && nc < 3; ++c) if (parent[c] == v) ch[nc++] = c;
    double dT, eT;
    if (nc == 2) {                                                         /* :199-216 */
        const DE L = bd_with(near, la, mu, rho, n_nodes, parent, len, ch[0]);
        const DE R = bd_with(near, la, mu, rho, n_nodes, parent, len, ch[1]);
        if (near) orp_compute_de_near_critical(la, mu, 1.0, br, L.E, &dT, &eT); else orp_compute_de(la, mu, 1.0, br, L.E, &dT, &eT);
        out.lnD = log(dT * la) + L.lnD + R.lnD;
        out.E = eT;
    } else if (nc == 1) {                                                  /* :218-223 */
        const DE C = bd_with(near, la, mu, rho, n_nodes, parent, len, ch[0]);
        if (near) orp_compute_de_near_critical(la, mu, 1.0, br, C.E, &dT, &eT); else orp_compute_de(la, mu, 1.0, br, C.E, &dT, &eT);
        out.lnD = log(dT * rho) + C.lnD;
        out.E = eT;
    } else if (nc == 0) {                                                  /* :225-231 */
        if (near) orp_compute_de_near_critical(la, mu, rho, br, 0.0, &dT, &eT); else orp_compute_de(la, mu, rho, br, 0.0, &dT, &eT);
        out.lnD = log(dT * rho);
        out.E = eT;
    } else {
        out.lnD = NAN; out.E = NAN;                                        /* "Tree is multifurcating." */
    }
    return out;
}

/* birthDeath cond la mu rho t -- BirthDeath.hs:158-184.  cond_mrca = 1: ConditionOnTimeOfMrca (product over
 * the two root sub trees), 0: ConditionOnTimeOfOrigin (the whole tree including its stem).
 * Returns NaN for structural faults (the reference calls `error`). */
double orp_birth_death(int cond_mrca, double la, double mu, double rho, int n_nodes, const int32_t *parent, const double *len)
{
    if (la < 0 || mu < 0 || rho <= 0 || rho > 1) return NAN;
    const int near = (1e-6 > fabs(la - mu));                               /* epsNearCritical */
    if (!cond_mrca) return bd_with(near, la, mu, rho, n_nodes, parent, len, 0).lnD;
    int ch[3], nc = 0;
    for (int c = 1; c < n_nodes && nc < 3; ++c) if (parent[c] == 0) ch[nc++] = c;
    if (nc != 2) return NAN;                                               /* "Tree is not bifurcating." */
    return bd_with(near, la, mu, rho, n_nodes, parent, len, ch[0]).lnD + bd_with(near, la, mu, rho, n_nodes, parent, len, ch[1]).lnD;
}

/* ---- relaxed molecular clock models ----------------------------------------------------------- */
/* logNormal' m v x -- RelaxedClock.hs:141-150 */
double orp_ln_lognormal_prime(double m, double v, double x)
{
    if (x <= 0) return -INFINITY;
    const double t = -(LN_SQRT_2_PI + log(x * sqrt(v)));
    const double a = 1.0 / (2 * v);
    const double b = log(x / m) + 0.5 * v;
    return t + (-(a * b * b));
}

enum { ORP_UNCORRELATED_GAMMA = 0, ORP_UNCORRELATED_LOGNORMAL = 1, ORP_UNCORRELATED_WHITE_NOISE = 2, ORP_AUTOCORRELATED_LOGNORMAL = 3 };

/* branchesWith WithoutStem f over the rate tree (Branch.hs:23-25) with the models of
 * RelaxedClock.hs:110-120, 160-166, 209-234, 307-324; mean m, variance v; tlen = time-tree branch lengths. */
double orp_relaxed_clock(int model, double m, double v, int n_nodes, const double *tlen, const double *rates)
{
    if (v <= 0) return NAN;                                                /* the reference calls `error` */
    double r = 0.0;
    for (int i = 1; i < n_nodes; ++i) {                                    /* WithoutStem */
        double k, th;
        switch (model) {
            case ORP_UNCORRELATED_GAMMA: shape_scale(m, v, &k, &th); r += orp_ln_gamma(k, th, rates[i]); break;
            case ORP_UNCORRELATED_LOGNORMAL: r += orp_ln_lognormal_prime(m, v, rates[i]); break;
            case ORP_UNCORRELATED_WHITE_NOISE: shape_scale(m, v / tlen[i], &k, &th); r += orp_ln_gamma(k, th, rates[i]); break;
            case ORP_AUTOCORRELATED_LOGNORMAL: r += orp_ln_lognormal_prime(m, v * tlen[i], rates[i]); break;
            default: return NAN;
        }
    }
    return r;
}

/* ---- priorFunction ht md cb cs bs x -- app/Probability.hs:127-150 ----------------------------- */
/* components[0..2] (optional): node priors, birth-death block, relaxed-clock block. */
double orp_prior(double ht, int model, int n_nodes, const int32_t *parent,
                 double birth, double death, double tH, const double *heights, double rMu, double rVar, const double *rates,
                 int ncal, const int32_t *cal_node, const int32_t *cal_has_lo, const double *cal_lo, const double *cal_lo_p,
                 const int32_t *cal_has_hi, const double *cal_hi, const double *cal_hi_p,
                 int ncon, const int32_t *con_young, const int32_t *con_old, const double *con_p,
                 int nbr, const int32_t *br_ptr, const int32_t *br_nodes, const double *br_sd, double *components)
{
    double *tlen = (double *)malloc(sizeof(double) * (size_t)n_nodes);
    for (int v = 0; v < n_nodes; ++v) tlen[v] = ((parent[v] < 0) ? heights[v] : heights[parent[v]]) - heights[v];   /* heightTreeToLengthTree */
    const double c0 = orp_node_priors(tH, heights, ncal, cal_node, cal_has_lo, cal_lo, cal_lo_p, cal_has_hi, cal_hi, cal_hi_p,
                                      ncon, con_young, con_old, con_p, nbr, br_ptr, br_nodes, br_sd);       /* :46-63 */
    const double c1 = orp_ln_exponential(1.0, birth) + orp_ln_exponential(1.0, death)
                      + orp_birth_death(1, birth, death, 1.0, n_nodes, parent, tlen);                        /* :66-85 */
    const double c2 = orp_ln_exponential(ht, rMu) + orp_ln_gamma(1.5, 1.0 / 6.0, rVar)
                      + orp_relaxed_clock(model, 1.0, rVar, n_nodes, tlen, rates);                           /* :96-124 */
    free(tlen);
    if (components) { components[0] = c0; components[1] = c1; components[2] = c2; }
    return c0 + c1 + c2;
}
