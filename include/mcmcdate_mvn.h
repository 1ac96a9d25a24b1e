/*
 * mcmcdate_mvn.h -- C ABI of the MI355X-native MVN phylogenetic log-likelihood.
 *
 * Drop-in boundary for the ONE hot path of dschrempf/mcmc-date: the likelihood closure that the
 * `mcmc` sampler calls once per proposal.  Every entry point names the reference interface it
 * replaces (paths relative to the reference repository).  Plain pointers and sizes only; no
 * C++/torch types.  The library is libmcmcdate_mvn.so (built from mcmc-date_amd/csrc/ by
 * __graft_entry__.build()).  INTEGRATION.md shows the Haskell `foreign import ccall` stubs.
 *
 * Conventions
 *   - All arithmetic is IEEE fp64.  Log-likelihoods are log-domain values (the argument of
 *     `Exp` in Numeric.Log), exactly what app/Probability.hs:169 computes.
 *   - Return value: MCD_OK (0) or a negative MCD_ERR_* for STRUCTURAL faults (bad sizes, non-SPD
 *     matrix, non-bifurcating root, HIP failure) -- the reference raises `error` for these
 *     (app/Tools.hs:43, app/Main.hs:220,231).  NUMERIC faults are not errors: NaN/Inf inputs
 *     flow through to NaN / -Inf outputs so that Metropolis-Hastings rejects the proposal, as
 *     in the reference (lib/Mcmc/Tree/Proposal/Unconstrained.hs:304-306).
 *   - mcd_last_error() returns a thread-local message for the last failing call of this thread.
 *   - Handles are immutable after creation; every evaluation entry point is re-entrant and may
 *     be called concurrently from several OS threads on the same handle (the reference runs the
 *     closure from several threads under `-threaded -N`, mcmc-date.cabal:42, app/Main.hs:452).
 *   - Batches are CHAIN-MAJOR: chain b's vector is contiguous at base + b * ld (ld >= length).
 *   - `on_device` = 0: all data pointers are host pointers; the call copies, runs and returns
 *     synchronously.  `on_device` = 1: all data pointers are device pointers on the handle's
 *     GPU, the kernels are enqueued on `stream` (a hipStream_t passed as void*, NULL = default
 *     stream) and the call returns without synchronising.
 *   - Trees are described by a PRE-ORDER parent array (root = 0, parent[0] = -1, every node
 *     before its children, children left to right): the order of elynx-tree's `branches` and of
 *     the reference's Foldable instances (lib/Mcmc/Tree/Types.hs:91-95, 146-150).
 */
#ifndef MCMCDATE_MVN_H
#define MCMCDATE_MVN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCD_OK 0
#define MCD_ERR_INVALID_ARG (-1)
#define MCD_ERR_NOT_SPD (-2)              /* app/Main.hs:220, 231: prepare aborts on such a matrix  */
#define MCD_ERR_HIP (-3)
#define MCD_ERR_ROOT_NOT_BIFURCATING (-4) /* app/Tools.hs:43 "getBranches: Root node is not bifurcating." */
#define MCD_ERR_NO_DEVICE (-5)
#define MCD_ERR_UNSUPPORTED (-6)

#define MCD_MAT_SIGMA 0     /* `mat` is the covariance matrix Sigma (output of meanCov, app/Main.hs:208) */
#define MCD_MAT_SIGMA_INV 1 /* `mat` is Sigma^-1 as stored in <name>.data (FullS, app/Main.hs:81, 240)    */

#define MCD_MAX_DIM 1024    /* largest MVN dimension this build holds in registers */

typedef struct mcd_mvn mcd_mvn_t;   /* replaces the closure `likelihoodFunction (Full mu s d)`  */
typedef struct mcd_tree mcd_tree_t; /* topology tables for the state -> distances wrapper        */

/* Number of usable GPUs (0 when there is none; never fails). */
int mcd_device_count(void);
const char* mcd_version(void);
const char* mcd_last_error(void);

/*
 * Build the immutable likelihood operands on GPU `device_id`.
 * Replaces: getLikelihoodFunction / getData (app/Main.hs:333-347, 85-99) + the closure creation
 * likelihoodFunction (Full mu sigmaInv logDetSigma) (app/Probability.hs:277-278, 247-248).
 *   n            MVN dimension (number of branches after merging the two root branches)
 *   mu           [n] posterior mean branch lengths
 *   mat          [n*n] row-major Sigma or Sigma^-1 (see mat_kind); only its symmetric part is used
 *   logdet_sigma log det Sigma; used as given when mat_kind = MCD_MAT_SIGMA_INV (the .data file
 *                carries it, app/Main.hs:240); ignored for MCD_MAT_SIGMA (computed from the factor)
 * The Cholesky factor L of Sigma is computed on the host and staged on the device once.
 */
int mcd_mvn_create(mcd_mvn_t** out, int n, const double* mu, const double* mat, int mat_kind,
                   double logdet_sigma, int device_id);
void mcd_mvn_destroy(mcd_mvn_t* h);
int mcd_mvn_dim(const mcd_mvn_t* h);
int mcd_mvn_device(const mcd_mvn_t* h);
double mcd_mvn_logdet(const mcd_mvn_t* h);
/* Copy the row-major lower Cholesky factor [n*n] of Sigma to host memory (diagnostics/tests). */
int mcd_mvn_get_factor(const mcd_mvn_t* h, double* L_out);

/*
 * One evaluation, host pointers.  Exact drop-in for
 *   logDensityFullMultivariateNormal mu (sigmaInvH, logDetSigma) xs      (app/Probability.hs:166-173)
 * x: [n]; *ll receives c - 1/2 (logdet + (x-mu)^T Sigma^-1 (x-mu)).
 */
int mcd_mvn_logpdf(const mcd_mvn_t* h, const double* x, double* ll);

/* `batch` independent evaluations of the same function (one per chain).  X: chain-major, ld >= n. */
int mcd_mvn_logpdf_batch(const mcd_mvn_t* h, const double* X, int64_t ld, int64_t batch, int on_device,
                         void* stream, double* ll);

/*
 * Log-likelihood and its gradient with respect to x:  G[b] = -Sigma^-1 (x_b - mu).
 * Replaces the AD of logDensityMultivariateNormalG / reduceVMV (app/Probability.hs:286-326)
 * that mcmc's NUTS performs per leapfrog step (app/Hamiltonian.hs:86-92).  G: chain-major, ldg >= n.
 */
int mcd_mvn_grad_batch(const mcd_mvn_t* h, const double* X, int64_t ld, int64_t batch, int on_device,
                       void* stream, double* ll, double* G, int64_t ldg);

/*
 * Bind a tree topology to a likelihood (n_nodes = n + 2; bifurcating root required).
 * Replaces the traversal structure implicit in getBranches / sumFirstTwo (app/Tools.hs:36-48)
 * and heightTreeToLengthTree (lib/Mcmc/Tree/Types.hs:224-233).
 */
int mcd_tree_create(mcd_tree_t** out, const mcd_mvn_t* h, int n_nodes, const int32_t* parent);
void mcd_tree_destroy(mcd_tree_t* t);
int mcd_tree_n_nodes(const mcd_tree_t* t);

/*
 * State -> log-likelihood, the closure the sampler holds:
 *   likelihoodFunctionWrapper logDensityFullMultivariateNormal mu (sigmaInv, logdet) x
 *                                                                    (app/Probability.hs:195-207)
 * Per chain b:  heights[b*ld_state + v] = relative node heights of x^.timeTree (leaves 0, root 1),
 *               rates[b*ld_state + v]   = branch labels of x^.rateTree (index 0 = stem, unused),
 *               tH[b] = x^.timeHeight,  rMu[b] = x^.rateMean          (app/State.hs:70-89).
 * log_jac (may be NULL) receives jacobianRootBranch x = log (1 / rootBranch x)
 *                                                                    (app/Probability.hs:393-410).
 */
int mcd_tree_loglik_batch(const mcd_tree_t* t, const double* heights, const double* rates, int64_t ld_state,
                          const double* tH, const double* rMu, int64_t batch, int on_device, void* stream,
                          double* ll, double* log_jac);

/*
 * Log-likelihood and its gradient with respect to the state, replacing the AD of
 * likelihoodFunctionG (app/Probability.hs:361-388) in htargetWith (app/Hamiltonian.hs:72-92).
 * g_heights / g_rates: [batch][ld_state] (every node; masking per app/Hamiltonian.hs:33-47 is the
 * caller's business; g_rates[.][0] = 0), g_tH / g_rMu: [batch].
 */
int mcd_tree_grad_batch(const mcd_tree_t* t, const double* heights, const double* rates, int64_t ld_state,
                        const double* tH, const double* rMu, int64_t batch, int on_device, void* stream,
                        double* ll, double* g_heights, double* g_rates, double* g_tH, double* g_rMu);

/* ------------------------------------------------------------------------------------------------
 * Prior (built after the likelihood rows): the batched log prior of the state.
 * Replaces  priorFunction ht md cb cs bs :: PriorFunction I  (app/Probability.hs:127-150), i.e.
 *   calibrateConstrainBraceSoft (lib/Mcmc/Tree/Prior/Node/Combined.hs:70-92)
 *   * exponential 1 birth * exponential 1 death * birthDeath ConditionOnTimeOfMrca birth death 1 t'
 *                                                                  (lib/Mcmc/Tree/Prior/BirthDeath.hs:158-239)
 *   * exponential ht rMu * gamma (3/2) (1/6) rVar * relaxed clock model 1 rVar t' rateTree
 *                                                                  (lib/Mcmc/Tree/Prior/Branch/RelaxedClock.hs)
 * Node indices are pre-order ids (the reference's `identify`, Calibration.hs:173).  Calibration boundaries are
 * absolute ages; cal_has_lo/hi = 0 encodes `Zero` / `Infinity`.  Braces: CSR (brace_ptr[n_brace+1], brace_nodes).
 * State-dependent faults for which the reference calls `error` (variance <= 0, negative birth/death rate)
 * come back as NaN for that chain; probability 0 is -Inf.
 * ---------------------------------------------------------------------------------------------- */
#define MCD_CLOCK_UNCORRELATED_GAMMA 0
#define MCD_CLOCK_UNCORRELATED_LOGNORMAL 1
#define MCD_CLOCK_UNCORRELATED_WHITE_NOISE 2
#define MCD_CLOCK_AUTOCORRELATED_LOGNORMAL 3

typedef struct mcd_prior mcd_prior_t;

int mcd_prior_create(mcd_prior_t** out, int n_nodes, const int32_t* parent, double ht, int clock_model,
                     int n_cal, const int32_t* cal_node, const int32_t* cal_has_lo, const double* cal_lo,
                     const double* cal_lo_p, const int32_t* cal_has_hi, const double* cal_hi, const double* cal_hi_p,
                     int n_con, const int32_t* con_young, const int32_t* con_old, const double* con_p,
                     int n_brace, const int32_t* brace_ptr, const int32_t* brace_nodes, const double* brace_sd,
                     int device_id);
void mcd_prior_destroy(mcd_prior_t* p);

/* Per chain b: birth[b], death[b], tH[b], heights[b*ld_state + v], rMu[b], rVar[b], rates[b*ld_state + v]
 * (the seven fields of `I`, app/State.hs:70-89).  lp[b] = log prior; components (may be NULL): [b][3] =
 * node priors, birth-death block, relaxed-clock block (what app/Monitor.hs:27-57 monitors). */
int mcd_prior_logprior_batch(const mcd_prior_t* p, const double* birth, const double* death, const double* tH,
                             const double* heights, const double* rMu, const double* rVar, const double* rates,
                             int64_t ld_state, int64_t batch, int on_device, void* stream, double* lp,
                             double* components);

#ifdef __cplusplus
}
#endif
#endif /* MCMCDATE_MVN_H */
