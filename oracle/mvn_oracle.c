/*
 * mvn_oracle.c -- CPU restatement of McmcDate's MVN phylogenetic log-likelihood path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under mcmc-date_amd/ (the product) may include,
 * link, import or execute this file.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py use it, and only as the checker / the timed baseline.
 *
 * PARITY UNPINNED: the reference (dschrempf/mcmc-date v1.0.0.0) ships no test suite,
 * no golden vectors and no expected outputs for this path (mcmc-date.cabal has no
 * test-suite stanza; tests/ holds input data only), and its Haskell toolchain is absent
 * from this image, so the reference itself cannot be run here.  This file therefore
 * restates the reference's formulas line by line (citations below) and is cross-checked
 * against an independent implementation (scipy.stats.multivariate_normal, see
 * tests/test_oracle.py) on operands derived from the reference's own tests/<NN>-leaves
 * input data.  The N^2 arithmetic of the reference is executed by hmatrix -> BLAS dgemv /
 * ddot (third-party, not vendored; version fixed only through stack.yaml:1 lts-21.22),
 * whose summation order is unspecified; parity is therefore to an fp64 tolerance.
 * (What the reference does commit are node ages SAMPLED with this likelihood -- the posterior runs of its mtCDNApri analysis.
 * Since round 3 the CPU twin built on THESE functions (oracle/mh_oracle.c) reproduces them without a GPU, all six node ages
 * within 0.5 % of the reference's means: tests/test_reference_samples.py::test_twin_reproduces_the_references_posterior_samples;
 * the device sampler likewise, tests/test_gpu_mh.py::test_posterior_node_ages_against_the_references_own_samples.  That pins
 * the path end to end, through a sampler -- a constant offset of ln likelihood would go unnoticed there, so single VALUES of this
 * file stay unpinned.)
 *
 * Tree representation used throughout: nodes are numbered in PRE-ORDER (root = 0, a node
 * before its children, children left to right), which is the order of elynx-tree's
 * `branches` / the Foldable instance used by the reference (lib/Mcmc/Tree/Types.hs:91-95,
 * 146-150).  `parent[v]` is the pre-order id of v's parent, parent[0] = -1.  Because of
 * pre-order numbering the children of v, left to right, are the nodes with parent v in
 * increasing id order.
 *
 * Build: see oracle/Makefile  (gcc -O2 -std=c11 -shared -fPIC, no -ffast-math).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* math-functions' m_ln_sqrt_2_pi, used at app/Probability.hs:173,184,193,326. */
#define ORC_LN_SQRT_2_PI 0.9189385332046727417803297364056176

#define ORC_OK 0
#define ORC_ERR_ROOT_NOT_BIFURCATING (-2) /* app/Tools.hs:43 `error` */
#define ORC_ERR_NOT_SPD (-3)
#define ORC_ERR_ARG (-1)

/* ------------------------------------------------------------------------------------
 * A1  logDensityFullMultivariateNormal -- app/Probability.hs:166-173
 *
 *   Exp $ c + (-0.5) * (logDetSigma + ((dxs <# sigmaInv) <.> dxs))
 *   dxs = xs - mu ; k = length mu ; c = negate (m_ln_sqrt_2_pi * k)
 *
 * `<#` is vector-times-matrix (y_k = sum_i dxs_i * sigmaInv[i][k]), `<.>` the dot product.
 * Returns the log-domain value (the argument of `Exp`).
 * ---------------------------------------------------------------------------------- */
double orc_logpdf_full(int n, const double *mu, const double *sigma_inv /* n x n row-major */,
                       double logdet_sigma, const double *xs)
{
    double *dxs = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *y = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) dxs[i] = xs[i] - mu[i];          /* :171 */
    for (int k = 0; k < n; ++k) y[k] = 0.0;
    for (int i = 0; i < n; ++i) {                                 /* dxs <# sigmaInv, :169 */
        const double *row = sigma_inv + (size_t)i * (size_t)n;
        const double di = dxs[i];
        for (int k = 0; k < n; ++k) y[k] += di * row[k];
    }
    double q = 0.0;
    for (int k = 0; k < n; ++k) q += y[k] * dxs[k];               /* <.> dxs, :169 */
    const double c = -(ORC_LN_SQRT_2_PI * (double)n);             /* :172-173 */
    free(dxs);
    free(y);
    return c + (-0.5) * (logdet_sigma + q);                       /* :169 */
}

/* Same formula in long double: the arbiter when two fp64 evaluation orders disagree. */
long double orc_logpdf_full_ld(int n, const double *mu, const double *sigma_inv,
                               double logdet_sigma, const double *xs)
{
    long double q = 0.0L;
    for (int k = 0; k < n; ++k) {
        long double yk = 0.0L;
        for (int i = 0; i < n; ++i)
            yk += ((long double)xs[i] - (long double)mu[i]) * (long double)sigma_inv[(size_t)i * n + k];
        q += yk * ((long double)xs[k] - (long double)mu[k]);
    }
    const long double c = -(0.9189385332046727417803297364056176L * (long double)n);
    return c + (-0.5L) * ((long double)logdet_sigma + q);
}

/* The quadratic form alone, (dxs <# sigmaInv) <.> dxs, same order as orc_logpdf_full. */
double orc_quadform_full(int n, const double *mu, const double *sigma_inv, const double *xs)
{
    return -2.0 * (orc_logpdf_full(n, mu, sigma_inv, 0.0, xs) + ORC_LN_SQRT_2_PI * (double)n);
}

/* ------------------------------------------------------------------------------------
 * logDensityUnivariateNormal -- app/Probability.hs:186-193, with
 * logSigmaSquaredProduct = sum (map log vs) from :274.
 * ---------------------------------------------------------------------------------- */
double orc_logpdf_univariate(int n, const double *mu, const double *vs, const double *xs)
{
    double lsp = 0.0, es = 0.0;
    for (int i = 0; i < n; ++i) lsp += log(vs[i]);                /* :274 */
    for (int i = 0; i < n; ++i) {                                 /* :190-191 */
        const double dx = xs[i] - mu[i];
        es += (dx * dx) / vs[i];
    }
    return -(ORC_LN_SQRT_2_PI * (double)n) + (-0.5) * (lsp + es); /* :188 */
}

/* logDensitySparseMultivariateNormal -- app/Probability.hs:178-184.
 * sigmaInvS as an association list (i, j, v) as stored in the .data file (app/Main.hs:79,
 * 142-155); dxs <.> (sigmaInvS !#> dxs). */
double orc_logpdf_sparse(int n, const double *mu, int64_t nnz, const int32_t *ii, const int32_t *jj,
                         const double *vv, double logdet_sigma, const double *xs)
{
    double *dxs = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *y = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    for (int i = 0; i < n; ++i) dxs[i] = xs[i] - mu[i];
    for (int64_t e = 0; e < nnz; ++e) y[ii[e]] += vv[e] * dxs[jj[e]];   /* !#> */
    double q = 0.0;
    for (int i = 0; i < n; ++i) q += dxs[i] * y[i];
    free(dxs);
    free(y);
    return -(ORC_LN_SQRT_2_PI * (double)n) + (-0.5) * (logdet_sigma + q);
}

/* ------------------------------------------------------------------------------------
 * Cholesky form asked for by BASELINE.json's north_star:
 *   q = || L^{-1} (x - mu) ||^2,  Sigma = L L^T,  logdet Sigma = 2 sum log L_ii.
 * Mathematically identical to A1; used to check that the two algebras agree on the
 * fixtures and as the model of what the HIP kernel computes.
 * ---------------------------------------------------------------------------------- */
int orc_cholesky(int n, const double *sigma /* row-major */, double *L /* row-major lower, out */)
{
    memset(L, 0, sizeof(double) * (size_t)n * (size_t)n);
    for (int j = 0; j < n; ++j) {
        long double s = sigma[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) s -= (long double)L[(size_t)j * n + k] * L[(size_t)j * n + k];
        if (!(s > 0.0L)) return ORC_ERR_NOT_SPD;
        const double ljj = (double)sqrtl(s);
        L[(size_t)j * n + j] = ljj;
        for (int i = j + 1; i < n; ++i) {
            long double t = sigma[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) t -= (long double)L[(size_t)i * n + k] * L[(size_t)j * n + k];
            L[(size_t)i * n + j] = (double)(t / ljj);
        }
    }
    return ORC_OK;
}

double orc_quadform_chol(int n, const double *mu, const double *L, const double *xs)
{
    double *z = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double q = 0.0;
    for (int i = 0; i < n; ++i) {
        double s = xs[i] - mu[i];
        const double *row = L + (size_t)i * n;
        for (int k = 0; k < i; ++k) s -= row[k] * z[k];
        z[i] = s / row[i];
        q += z[i] * z[i];
    }
    free(z);
    return q;
}

double orc_logpdf_chol(int n, const double *mu, const double *L, const double *xs)
{
    double logdet = 0.0;
    for (int i = 0; i < n; ++i) logdet += 2.0 * log(L[(size_t)i * n + i]);
    return -(ORC_LN_SQRT_2_PI * (double)n) + (-0.5) * (logdet + orc_quadform_chol(n, mu, L, xs));
}

/* d ll / d xs for A1.  The reference obtains gradients by automatic differentiation of
 * logDensityMultivariateNormalG (app/Probability.hs:309-326, reduceVMV :286-298); the exact
 * derivative of  -1/2 * sum_ij dx_i P_ij dx_j  is  -1/2 (P + P^T) dx, restated here. */
void orc_grad_full(int n, const double *mu, const double *sigma_inv, const double *xs, double *g)
{
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j)
            s += (sigma_inv[(size_t)i * n + j] + sigma_inv[(size_t)j * n + i]) * (xs[j] - mu[j]);
        g[i] = -0.5 * s;
    }
}

/* ------------------------------------------------------------------------------------
 * A4  heightTreeToLengthTree -- lib/Mcmc/Tree/Types.hs:224-233
 *   go hParent (Node hNode ..) = Node (hParent - hNode) .. (map (go hNode) ts), started with
 *   go (branch t) t, i.e. the root gets hRoot - hRoot.  No negativity check (:229-231).
 * ---------------------------------------------------------------------------------- */
void orc_height_to_length(int n_nodes, const int32_t *parent, const double *heights, double *lengths)
{
    for (int v = 0; v < n_nodes; ++v) {
        const double hp = (parent[v] < 0) ? heights[v] : heights[parent[v]];
        lengths[v] = hp - heights[v];
    }
}

/* size of the subtree rooted at v (pre-order ids: subtree(v) = [v, v + size)). */
static int subtree_size(int n_nodes, const int32_t *parent, int v)
{
    int e = v + 1;
    while (e < n_nodes) {
        /* e belongs to subtree(v) iff walking up from e reaches v before passing below v */
        int a = e;
        while (a > v) a = parent[a];
        if (a != v) break;
        ++e;
    }
    return e - v;
}

/* ------------------------------------------------------------------------------------
 * A3  getBranches -- app/Tools.hs:36-43
 *   getBranches (Node _ _ [l, r]) = fromList $ head ls : head rs : tail ls ++ tail rs
 *     where ls = branches l ; rs = branches r        (pre-order branch labels)
 *   getBranches _ = error "getBranches: Root node is not bifurcating."
 * `values[v]` is the branch label of pre-order node v.  Output length n_nodes - 1.
 * ---------------------------------------------------------------------------------- */
int orc_get_branches(int n_nodes, const int32_t *parent, const double *values, double *out)
{
    int nroot = 0, l = -1, r = -1;
    for (int v = 1; v < n_nodes; ++v)
        if (parent[v] == 0) {
            if (nroot == 0) l = v; else if (nroot == 1) r = v;
            ++nroot;
        }
    if (n_nodes < 3 || nroot != 2) return ORC_ERR_ROOT_NOT_BIFURCATING;
    const int sl = subtree_size(n_nodes, parent, l);
    const int sr = subtree_size(n_nodes, parent, r);
    if (l != 1 || r != 1 + sl || 1 + sl + sr != n_nodes) return ORC_ERR_ARG;
    int o = 0;
    out[o++] = values[l];                                   /* head ls */
    out[o++] = values[r];                                   /* head rs */
    for (int v = l + 1; v < l + sl; ++v) out[o++] = values[v]; /* tail ls */
    for (int v = r + 1; v < r + sr; ++v) out[o++] = values[v]; /* tail rs */
    return ORC_OK;
}

/* sumFirstTwo -- app/Tools.hs:47-48:  (v!0 + v!1) `cons` drop 2 v.   len_in >= 2. */
void orc_sum_first_two(int len_in, const double *v, double *out)
{
    out[0] = v[0] + v[1];
    for (int i = 2; i < len_in; ++i) out[i - 1] = v[i];
}

/* ------------------------------------------------------------------------------------
 * A2  likelihoodFunctionWrapper -- app/Probability.hs:195-207
 *   times = getBranches (heightTreeToLengthTree (x ^. timeTree))
 *   rates = getBranches (x ^. rateTree)
 *   distances = map (* (tH * rMu)) $ sumFirstTwo $ zipWith (*) times rates
 * heights[v], rates[v] indexed by pre-order node (rates[0] = stem, unused).
 * distances has n_nodes - 2 entries.
 * ---------------------------------------------------------------------------------- */
int orc_distances(int n_nodes, const int32_t *parent, const double *heights, const double *rates,
                  double tH, double rMu, double *distances)
{
    double *len = (double *)malloc(sizeof(double) * (size_t)n_nodes);
    double *tb = (double *)malloc(sizeof(double) * (size_t)n_nodes);
    double *rb = (double *)malloc(sizeof(double) * (size_t)n_nodes);
    orc_height_to_length(n_nodes, parent, heights, len);
    int rc = orc_get_branches(n_nodes, parent, len, tb);                 /* :203 */
    if (rc == ORC_OK) rc = orc_get_branches(n_nodes, parent, rates, rb); /* :204 */
    if (rc == ORC_OK) {
        const int nb = n_nodes - 1;
        for (int i = 0; i < nb; ++i) tb[i] = tb[i] * rb[i];              /* zipWith (*) */
        orc_sum_first_two(nb, tb, distances);                            /* sumFirstTwo */
        const double s = tH * rMu;                                       /* :207 */
        for (int i = 0; i < nb - 1; ++i) distances[i] = distances[i] * s;
    }
    free(len);
    free(tb);
    free(rb);
    return rc;
}

/* likelihoodFunction (Full mu s d) applied to a state -- app/Probability.hs:277-278, 247-248. */
int orc_tree_loglik_full(int n_nodes, const int32_t *parent, const double *heights, const double *rates,
                         double tH, double rMu, const double *mu, const double *sigma_inv,
                         double logdet_sigma, double *ll)
{
    const int n = n_nodes - 2;
    double *d = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int rc = orc_distances(n_nodes, parent, heights, rates, tH, rMu, d);
    if (rc == ORC_OK) *ll = orc_logpdf_full(n, mu, sigma_inv, logdet_sigma, d);
    free(d);
    return rc;
}

/* ------------------------------------------------------------------------------------
 * A8  rootBranch / jacobianRootBranch -- app/Probability.hs:393-410
 *   rootBranch x = tH * rM * (t1 * r1 + t2 * r2);  jacobian = Exp . log . recip . rootBranch
 * Returns log(1 / rootBranch) (the log-domain value), NaN-propagating.
 * ---------------------------------------------------------------------------------- */
int orc_log_jacobian_root_branch(int n_nodes, const int32_t *parent, const double *heights,
                                 const double *rates, double tH, double rMu, double *logj)
{
    int nroot = 0, l = -1, r = -1;
    for (int v = 1; v < n_nodes; ++v)
        if (parent[v] == 0) {
            if (nroot == 0) l = v; else if (nroot == 1) r = v;
            ++nroot;
        }
    if (nroot != 2) return ORC_ERR_ROOT_NOT_BIFURCATING;                 /* :398, :401 */
    const double t1 = heights[0] - heights[l], t2 = heights[0] - heights[r];
    const double rb = tH * rMu * (t1 * rates[l] + t2 * rates[r]);        /* :394 */
    *logj = log(1.0 / rb);                                               /* :409 */
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------
 * A7  gradient of likelihoodFunctionG (app/Probability.hs:361-388) with respect to the
 * state, obtained in the reference by AD inside mcmc's NUTS (app/Hamiltonian.hs:86-92).
 * Restated analytically (chain rule through A2/A3/A4):
 *   g = d ll / d distances = -1/2 (P + P^T)(d - mu),       s = tH * rMu
 *   row(v) = slot of node v in the distance vector (both root children -> slot 0)
 *   d ll / d rate[v]    = s * g[row(v)] * t_v                      (v != root)
 *   d ll / d height[v] += -s * g[row(v)] * rate[v]                 (v != root)
 *   d ll / d height[p] += +s * g[row(v)] * rate[v]   p = parent(v)
 *   d ll / d tH  = (g . d) / tH ;  d ll / d rMu = (g . d) / rMu
 * Gradients with respect to ALL heights are returned (root and leaves included); masking
 * (app/Hamiltonian.hs:33-47) is the caller's business.
 * ---------------------------------------------------------------------------------- */
int orc_tree_grad_full(int n_nodes, const int32_t *parent, const double *heights, const double *rates,
                       double tH, double rMu, const double *mu, const double *sigma_inv,
                       double *g_heights, double *g_rates, double *g_tH, double *g_rMu)
{
    const int n = n_nodes - 2;
    double *d = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *g = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *slotf = (double *)malloc(sizeof(double) * (size_t)n_nodes);
    double *ids = (double *)malloc(sizeof(double) * (size_t)n_nodes);
    int rc = orc_distances(n_nodes, parent, heights, rates, tH, rMu, d);
    if (rc == ORC_OK) {
        /* slot of every node: run getBranches on the identity labelling. */
        for (int v = 0; v < n_nodes; ++v) ids[v] = (double)v;
        orc_get_branches(n_nodes, parent, ids, slotf);
        int *row = (int *)malloc(sizeof(int) * (size_t)n_nodes);
        row[0] = -1;
        for (int i = 0; i < n_nodes - 1; ++i) row[(int)slotf[i]] = (i < 2) ? 0 : i - 1;
        orc_grad_full(n, mu, sigma_inv, d, g);
        const double s = tH * rMu;
        for (int v = 0; v < n_nodes; ++v) g_heights[v] = 0.0, g_rates[v] = 0.0;
        for (int v = 1; v < n_nodes; ++v) {
            const int p = parent[v];
            const double t = heights[p] - heights[v];
            g_rates[v] = s * g[row[v]] * t;
            g_heights[v] -= s * g[row[v]] * rates[v];
            g_heights[p] += s * g[row[v]] * rates[v];
        }
        double gd = 0.0;
        for (int i = 0; i < n; ++i) gd += g[i] * d[i];
        *g_tH = gd / tH;
        *g_rMu = gd / rMu;
        free(row);
    }
    free(d);
    free(g);
    free(slotf);
    free(ids);
    return rc;
}

/* ------------------------------------------------------------------------------------
 * Batch drivers: one evaluation per chain, exactly the reference's call pattern (one
 * closure call per proposal, fresh dxs buffer per call).  X is chain-major: chain b's
 * vector starts at X + b * ld.  Used by tests and by bench.py's cpu_baseline ("port").
 * ---------------------------------------------------------------------------------- */
void orc_logpdf_full_batch(int n, const double *mu, const double *sigma_inv, double logdet_sigma,
                           const double *X, int64_t ld, int64_t batch, double *ll)
{
    for (int64_t b = 0; b < batch; ++b) ll[b] = orc_logpdf_full(n, mu, sigma_inv, logdet_sigma, X + b * ld);
}

/* the same, chains split statically over the host's cores (bench.py cpu_baseline, all-core figure) */
void orc_logpdf_full_batch_mt(int n, const double *mu, const double *sigma_inv, double logdet_sigma,
                              const double *X, int64_t ld, int64_t batch, double *ll)
{
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < batch; ++b) ll[b] = orc_logpdf_full(n, mu, sigma_inv, logdet_sigma, X + b * ld);
}

void orc_logpdf_chol_batch(int n, const double *mu, const double *L, const double *X, int64_t ld,
                           int64_t batch, double *ll)
{
    for (int64_t b = 0; b < batch; ++b) ll[b] = orc_logpdf_chol(n, mu, L, X + b * ld);
}

void orc_grad_full_batch(int n, const double *mu, const double *sigma_inv, const double *X, int64_t ld,
                         int64_t batch, double *G, int64_t ldg)
{
    for (int64_t b = 0; b < batch; ++b) orc_grad_full(n, mu, sigma_inv, X + b * ld, G + b * ldg);
}

int orc_tree_loglik_full_batch(int n_nodes, const int32_t *parent, const double *heights /* [batch][n_nodes] */,
                               const double *rates /* [batch][n_nodes] */, const double *tH, const double *rMu,
                               const double *mu, const double *sigma_inv, double logdet_sigma, int64_t batch,
                               double *ll, double *log_jac)
{
    for (int64_t b = 0; b < batch; ++b) {
        int rc = orc_tree_loglik_full(n_nodes, parent, heights + b * n_nodes, rates + b * n_nodes, tH[b],
                                      rMu[b], mu, sigma_inv, logdet_sigma, ll + b);
        if (rc != ORC_OK) return rc;
        if (log_jac) {
            rc = orc_log_jacobian_root_branch(n_nodes, parent, heights + b * n_nodes, rates + b * n_nodes,
                                              tH[b], rMu[b], log_jac + b);
            if (rc != ORC_OK) return rc;
        }
    }
    return ORC_OK;
}
