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
 * Which form of the log-density kernels a launch uses (results agree to rounding, not bit for bit):
 *   MCD_FORM_AUTO     (default) by dimension and batch size: the column sweep up to N = 256 at a sampler's usual batch; for
 *                     N > 256 and up to 1024 chains (240 < N <= 256: up to 128) a row-split variant of the multiply form
 *                     (k_split.hip: the row blocks of L^-1 dealt to 8-32 workgroups per 16-chain tile, partial sums handed over
 *                     through a per-stream scratch owned by the handle), raw x and tree states alike; the multiply form from
 *                     2048 chains at N >= 96, from 8192 at N >= 32;
 *   MCD_FORM_SWEEP    always the column sweep (one chain per wave -- the latency form);
 *   MCD_FORM_MULTIPLY always the multiply form z = L^-1 (x - mu) on the fp64 matrix cores (the throughput form).
 * mcd_mvn_set_form chooses per handle (two samplers with a handle each can pin different forms); a handle whose choice is
 * MCD_FORM_AUTO follows the process default, which mcd_set_logpdf_form sets (a test and tuning knob; the environment
 * variable MCD_WIDE=0|1, read when the library is loaded, sets its initial value to SWEEP | MULTIPLY).  Both return the previous value.
 * The gradient entry points follow the same choice; above N = 256 the multiply form uses the gradient rows (`G`, `g_heights`)
 * as scratch while it runs: a call whose gradient array IS an input array (in place) takes the sweeps instead; partially
 * overlapping arrays are not supported by either form.  Above N = 768 the gradients have no sweep form (sixteen row blocks of both
 * sweeps' staging do not fit the register file): whatever the batch and the form asked for they take the row-split form in pieces
 * of at most 1024 chains; mcd_mvn_grad_batch may still be called in place, mcd_tree_grad_batch may write g_heights over heights and
 * g_rates over rates, but an output over the OTHER input array is refused there (MCD_ERR_HIP / invalid value).
 * Stream capture: every entry point that takes a stream may be captured into a hipGraph.  The row-split form gives a stream
 * under capture a scratch set of the capture's own, so a graph may be replayed on any stream while eager calls go on on the
 * stream it was captured from; two executable graphs instantiated from ONE capture must not be replayed concurrently.
 * A scratch set (256 KiB; gradient calls add two arrays of 1024 chains x 16 ceil(N/16) doubles, allocated at the first gradient
 * call) stays with its stream until the handle is destroyed.  A host that makes short-lived streams calls
 * mcd_mvn_release_stream(h, stream) before destroying one: it waits for the stream and returns its set to the handle's pool
 * (sets of captures stay with the capture: its graph may still be replayed).  hipStreamPerThread is keyed per calling thread.
 */
/*
 * Test and tuning knobs -- ONE explicit, thread-safe table instead of environment variables read on the hot path (rounds 1-3 called
 * getenv() per launch: a stray variable silently changed the launch structure and the rounding of a run, and getenv() races with setenv()
 * in a threaded host such as the reference's, mcmc-date.cabal:42-43).  name = "MCD_MH_SEGMENTS", "MCD_MH_INCREMENTAL", "MCD_MH_PER_PHASE",
 * "MCD_MH_PRIOR", "MCD_MH_PRIOR_CACHE", "MCD_MH_STEP_WG", "MCD_MH_CHAIN_LW", "MCD_MH_INC_SLOTS", "MCD_MH_SPARSE_SLOTS", "MCD_SPLIT", "MCD_SPLIT_G",
 * "MCD_SPLIT_SCATTER", "MCD_SPLIT_NOROT", "MCD_SPLIT_PROBE", "MCD_GEOM", "MCD_WIDE_CT", "MCD_SPARSE_QUAD", "MCD_MH_PRIOR_WAVES", "MCD_LOADERS",
 * "MCD_MH_SEG_TAIL" (0: a dense proposal after a segment is proposed by the step kernel), "MCD_MH_AHEAD_FROM" (nodes from which a segment's chain
 * wave draws the next proposal ahead of the decision), "MCD_MH_PRIOR_DRAWS" (0: no prior wave draws the next proposal's rejected branch); value = a decimal integer, NULL or ""
 * = back to the default.  What each knob does is said where it acts (mcd_mh_run, the forms above).  The environment variables of the same
 * names are read ONCE, when the library is loaded, as initial values -- never afterwards.  No knob changes a result beyond rounding.
 */
int mcd_set_option(const char* name, const char* value);
int mcd_get_option(const char* name, int* is_set, int* value);

#define MCD_FORM_AUTO 0
#define MCD_FORM_SWEEP 1
#define MCD_FORM_MULTIPLY 2
int mcd_set_logpdf_form(int form);
int mcd_mvn_set_form(const mcd_mvn_t* h, int form);
int mcd_mvn_release_stream(const mcd_mvn_t* h, void* stream /* hipStream_t */);

/*
 * Build the immutable likelihood operands on GPU `device_id`.
 * Replaces: getLikelihoodFunction / getData (app/Main.hs:333-347, 85-99) + the closure creation
 * likelihoodFunction (Full mu sigmaInv logDetSigma) (app/Probability.hs:277-278, 247-248).
 *   n            MVN dimension (number of branches after merging the two root branches)
 *   mu           [n] posterior mean branch lengths
 *   mat          [n*n] row-major Sigma or Sigma^-1 (see mat_kind); only its symmetric part is used
 *   logdet_sigma log det Sigma; used as given when mat_kind = MCD_MAT_SIGMA_INV (the .data file
 *                carries it, app/Main.hs:240); ignored for MCD_MAT_SIGMA (computed from the factor)
 * The factors are computed on the host and staged on the device once: from Sigma its Cholesky factor; from Sigma^-1 = P the
 * factor W with P = W^T W directly (a reverse Cholesky factorisation: nothing is inverted twice) and L = W^-1 for the sweeps.
 * Contract, stricter than the reference's: the matrix must be numerically positive definite -- MCD_ERR_NOT_SPD otherwise.  (The
 * reference evaluates dx^T P dx with whatever P the .data file holds; `prepare` itself refuses a covariance matrix whose
 * determinant is not positive, app/Main.hs:231, so an indefinite P only arises from a hand-made file.  For such a matrix use the
 * product form, mcd_sparse_create below, which takes P as given -- the host mirrors' likelihoodFunction do so by themselves.)
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

/*
 * Log prior and its gradient with respect to the seven fields of the state (first part): the prior
 * factor of the Hamiltonian target `htargetWith` (app/Hamiltonian.hs:72-92), which the reference differentiates by AD.
 * g_heights / g_rates: [batch][ld_state] (every node; masking per app/Hamiltonian.hs:33-47 is the caller's business;
 * g_rates[.][0] = 0).  Outside the support (ln prior = -inf or NaN) every gradient entry of that chain is NaN.  In the
 * near-critical regime of the birth-death prior (|birth - death| < 1e-6, BirthDeath.hs:117-118) the gradient is the one
 * of the exact formulas at the edge of that regime (relative deviation O(1e-6) from the first-order value's derivative).
 */
int mcd_prior_grad_batch(const mcd_prior_t* p, const double* birth, const double* death, const double* tH,
                         const double* heights, const double* rMu, const double* rVar, const double* rates,
                         int64_t ld_state, int64_t batch, int on_device, void* stream, double* lp, double* g_birth,
                         double* g_death, double* g_tH, double* g_heights, double* g_rMu, double* g_rVar, double* g_rates);

/*
 * Leapfrog integrator of the Hamiltonian proposal on the device (second part).  Potential
 * U(q) = -ln [prior x likelihood x jacobianRootBranch] (`htargetWith`, app/Hamiltonian.hs:72-92); the position q is the
 * masked, reversed fold of the state (`getMask`, `toVector`, :33-53): root height, leaf heights and the rate stem are
 * fixed, the time height moves only when calibrations are available.  The state of `batch` chains lives on the device;
 *   mcd_hmc_set_state      host state in, evaluates ln target and its gradient
 *   mcd_hmc_get_position   q [batch][dim], ln target [batch], gradient [batch][dim] of the current state (any may be NULL)
 *   mcd_hmc_leapfrog       n_steps leapfrog steps with per-chain step size eps[b], per-chain direction dir[b] = +-1 (NULL
 *                          = forward) and diagonal inverse masses inv_mass[dim]; p [batch][dim] in/out (host)
 * A state that leaves the support gets NaN gradients; its momentum and ln target turn NaN (the caller rejects).
 *
 * The proposal itself -- `nuts` of the package `mcmc` (`nutsWith`, app/Hamiltonian.hs:95-105; not vendored), restated from the
 * algorithm it implements, Hoffman & Gelman (2014), Algorithm 3 -- runs on the device for all chains in lock step (k_nuts.hip:
 * slice variable, doubling in a random direction, U-turn and divergence stops, uniform choice among the admissible leaves;
 * one round = the two gradient launches + one launch that finishes the leapfrog step, books the leaf and sends the next one
 * on its way; the host only polls a counter):
 *   mcd_hmc_nuts        one transition from the handle's state with per-chain step sizes eps[batch] and inverse masses
 *                       inv_mass[dim]; random streams: Philox (seed, chain_offset + b, transition) -- a chain's draw does not
 *                       depend on the batch it runs in; out: mean acceptance statistic alpha[batch], tree depth[batch].
 *                       The streams share the counter layout of mcd_mh_*: a host that runs both on one seed passes
 *                       seed ^ constant here (the Python mirror: hmc.NUTS_STREAM_DOMAIN) and a transition counter that never
 *                       restarts.  After the call the handle holds the accepted point with its value and gradient, so any
 *                       entry point (mcd_hmc_leapfrog, mcd_hmc_get_position, another transition) may follow directly.
 *   mcd_hmc_nuts_run    n transitions; adapt != 0: dual averaging of the step sizes towards the acceptance statistic delta
 *                       (Algorithm 6; eps in: starting value, out: the averaged value); mean_alpha[batch]; q_mean / q_var[dim]:
 *                       position means and variances pooled over chains and transitions (what a mass adaptation needs:
 *                       the reference tunes `HTuneLeapfrog HTuneAllMasses`, app/Hamiltonian.hs:62-63).
 * `mcmc`'s own defaults and tuning schedule are not available here: parity of this part rests on the CPU twin
 * (tests/test_gpu_nuts.py) and on the agreement with Metropolis-Hastings chains.
 */
typedef struct mcd_hmc mcd_hmc_t;
int mcd_hmc_create(mcd_hmc_t** out, const mcd_tree_t* tree, const mcd_prior_t* prior, int calibrations_available, int64_t batch);
void mcd_hmc_destroy(mcd_hmc_t* m);
int mcd_hmc_dim(const mcd_hmc_t* m);
int mcd_hmc_set_state(mcd_hmc_t* m, const double* birth, const double* death, const double* tH, const double* heights,
                      const double* rMu, const double* rVar, const double* rates, int64_t ld_state);
int mcd_hmc_get_state(const mcd_hmc_t* m, double* birth, double* death, double* tH, double* heights, double* rMu,
                      double* rVar, double* rates, int64_t ld_state);
int mcd_hmc_get_position(const mcd_hmc_t* m, double* q, double* value, double* grad);
int mcd_hmc_leapfrog(mcd_hmc_t* m, double* p, const double* eps, const double* dir, const double* inv_mass, int n_steps);
int mcd_hmc_nuts(mcd_hmc_t* m, const double* eps, const double* inv_mass, int max_depth, uint64_t seed, int64_t chain_offset,
                 uint64_t transition, double* alpha, int32_t* depth);
int mcd_hmc_nuts_run(mcd_hmc_t* m, int n_transitions, int adapt, double* eps, const double* inv_mass, double delta, int max_depth,
                     uint64_t seed, int64_t chain_offset, uint64_t first_transition, double* mean_alpha, double* q_mean, double* q_var);
/* Step sizes AND masses (`HTuneLeapfrog HTuneAllMasses`, app/Hamiltonian.hs:62-63): `windows` windows of `window` transitions --
 * dual averaging of eps[batch] inside a window, then inv_mass[dim] := pooled variance of the positions the window visited
 * (shrunk towards 1e-3) -- and a closing window for the step sizes; uses the transitions first_transition ..
 * first_transition + (windows + 1) window - 1 of the random streams.  eps, inv_mass: in = starting values, out = tuned. */
int mcd_hmc_nuts_warmup(mcd_hmc_t* m, int windows, int window, double* eps, double* inv_mass, double delta, int max_depth, uint64_t seed,
                        int64_t chain_offset, uint64_t first_transition, double* mean_alpha);
/* One leapfrog step from ARBITRARY phase points (what a NUTS tree needs: it extends either end of a trajectory):
 * q, p, grad [batch][dim] in/out (host), value [batch] out (ln target at the new point, may be NULL).  have_grad = 0:
 * the gradient at q is evaluated first (grad is output only).  The handle's own state becomes the new point. */
int mcd_hmc_step_from(mcd_hmc_t* m, double* q, double* p, double* grad, int have_grad, const double* eps, const double* dir,
                      const double* inv_mass, double* value);

/* ------------------------------------------------------------------------------------------------
 * Batched Metropolis-Hastings-Green driver (FIRST SLICE).  `mcmc`'s `mhg` evaluates one state
 * per call (app/Main.hs:474); here `batch` independent chains execute the same proposal of the cycle at the same
 * time, each with its own random numbers, tuning parameter and accept/reject decision.  The state never leaves
 * the device between calls.  A proposal is a row of the table below; the caller builds the table and the
 * per-iteration order (the reference: app/Definitions.hs:127-278 `proposals`, weights replicated and shuffled by
 * `mcmc`).  Every proposal of the reference cycle is a kind below; NUTS (app/Hamiltonian.hs, row f3) is not built.
 * ---------------------------------------------------------------------------------------------- */
#define MCD_PROP_SCALE_SCALAR 0        /* scaleUnbiased k [mcmc]: node = 0 birth, 1 death, 2 tH, 3 rMu, 4 rVar; p0 = k          */
#define MCD_PROP_SLIDE_NODE 1          /* slideNodeAtUltrametric (Ultrametric.hs:50-59): node; p0 = sd                          */
#define MCD_PROP_SCALE_SUBTREE_TIME 2  /* scaleSubTreeAtUltrametric (:126-149): node; p0 = sd; n1 = inner nodes of the sub tree  */
#define MCD_PROP_PULLEY 3              /* pulleyUltrametric (:221-286): p0 = sd; n1, n2 = inner nodes left / right               */
#define MCD_PROP_SCALE_BRANCH_RATE 4   /* scaleBranch (Unconstrained.hs:40-66): node; p0 = shape                                 */
#define MCD_PROP_SCALE_SUBTREE_RATE 5  /* scaleTree on a sub tree (:84-130): node; p0 = shape; n1 = nodes of the sub tree        */
#define MCD_PROP_SCALE_NORM_TREE 6     /* scaleNormAndTreeContrarily (:221-256): node = 2 (tH) or 3 (rMu); p0 = shape            */
#define MCD_PROP_SCALE_VAR_TREE 7      /* scaleVarianceAndTree (:286-316): p0 = shape; p1 = 1 uses the determinant u^(n-1) as
                                          Jacobian instead of the reference's (u - u/n + 1/n)^n            */
#define MCD_PROP_SCALE_VAR_TREE_AUTO 8 /* scaleVarianceAndTreeAutocorrelated (:354-386): p0 = shape                              */
#define MCD_PROP_SCALE_CONTRARILY 9    /* scaleContrarily k th [mcmc] on (tH, rMu): p0 = k, p1 = th                              */
#define MCD_PROP_SLIDE_NODE_CONTRA 10  /* slideNodesAtContrarily (Contrary.hs:35-77): node; p0 = sd                              */
#define MCD_PROP_SCALE_SUBTREE_CONTRA 11 /* scaleSubTreesAtContrarily (:269-326): node; p0 = sd; n1 = inner nodes, n2 = nodes    */
#define MCD_PROP_SLIDE_ROOT_CONTRA 12  /* slideRootContrarily (:191-223) on (tH, time tree, rate tree): p0 = sd; n1 = exponent of
                                          1/u in the Jacobian: the reference passes the number of inner nodes INCLUDING the root */
#define MCD_PROP_SCALE_RATES_TREE_CONTRA 13 /* scaleRatesAndTreeContrarily (:420-446) on (birth rate, rate mean, time tree):
                                          p0 = sd; n1 = inner nodes - 1                                                         */
#define MCD_PROP_SLIDE_BRACE 14        /* slideBracedNodesUltrametric (Brace.hs:98-156): node = brace index of the prior; p0 = sd */
#define MCD_PROP_SLIDE_BRACE_CONTRA 15 /* slideBracedNodesContrarily (Brace.hs:37-61): node = brace index; p0 = sd                */

typedef struct mcd_mh mcd_mh_t;

/*
 * tree, prior: handles on the same device; they must outlive the driver.  Proposal table, n_prop rows: kind, node,
 * n1, n2 (see above), jac_root = 1 for proposals lifted with jacobianRootBranch (the "[R]" proposals of
 * app/Definitions.hs:145-278), dim = PDimension (selects the optimal acceptance rate of the auto tuner), p0, p1.
 * All chains start with tuning parameter 1.  Random numbers are Philox4x32-10 keyed by `seed`, counter =
 * (draw, chain, step): results do not depend on batch size, launch geometry or GPU count.
 * Unlike the likelihood and prior handles, a driver handle (mcd_mh_t, mcd_hmc_t) carries the chains' state: use it from
 * one thread at a time; different handles may be driven concurrently from different threads.
 */
int mcd_mh_create(mcd_mh_t** out, const mcd_tree_t* tree, const mcd_prior_t* prior, int n_prop, const int32_t* kind,
                  const int32_t* node, const int32_t* n1, const int32_t* n2, const int32_t* jac_root, const int32_t* dim,
                  const double* p0, const double* p1, int64_t batch, uint64_t seed);
/* The same driver over a likelihood whose precision matrix stays sparse on the device (mcd_sparse_tree_create): the reference's PRODUCTION
 * configuration -- `mhg` with likelihoodFunction (Sparse ...) (app/Main.hs:474, 257-277, 333-347; app/Probability.hs:178-184, 279); every
 * published timing of the reference uses it.  Trees of 3 .. 2048 nodes (a Sparse record never has to be densified).  mcd_mh_run takes the
 * segment structure (MCD_MH_PATH_SEGMENTS_SPARSE): every run of steps between two proposals that move more than 254 distances in ONE launch,
 * the chains' states in LDS, the quadratic form q = dx^T P dx kept per chain and updated through the ROWS of the moved distances,
 *     q' = q + sum_{j moved} delta_j sum_k Ps[j][k] (2 dx_k + delta_k),   Ps = (P + P^T) / 2;
 * such a dense proposal by two launches (the step kernel proposes, the one-launch full form evaluates), q recomputed by a full form every
 * 256 steps.  On a tree whose distances all fit the list (at most 256 nodes) every proposal runs inside a segment.  MCD_MH_SEGMENTS=0 or
 * MCD_MH_INCREMENTAL=0 (mcd_set_option): the step kernel + a full product at every step (round 3's structure; the chains: same decisions and
 * states).  Everything else (mcd_mh_set_state, _run, _tune, _mc3_*, ...) as for mcd_mh_create.  The mcd_sparse_tree_t type is declared with
 * the sparse form below. */
struct mcd_sparse_tree;
int mcd_mh_create_sparse(mcd_mh_t** out, const struct mcd_sparse_tree* tree, const mcd_prior_t* prior, int n_prop, const int32_t* kind,
                         const int32_t* node, const int32_t* n1, const int32_t* n2, const int32_t* jac_root, const int32_t* dim,
                         const double* p0, const double* p1, int64_t batch, uint64_t seed);
void mcd_mh_destroy(mcd_mh_t* m);
/* first_chain: global index of this handle's chain 0 (chain shards on several GPUs draw disjoint random streams). */
int mcd_mh_set_chain_offset(mcd_mh_t* m, int64_t first_chain);
/* Host arrays, the seven fields of `I` per chain (as mcd_prior_logprior_batch); evaluates the posterior. */
int mcd_mh_set_state(mcd_mh_t* m, const double* birth, const double* death, const double* tH, const double* heights,
                     const double* rMu, const double* rVar, const double* rates, int64_t ld_state);
int mcd_mh_get_state(const mcd_mh_t* m, double* birth, double* death, double* tH, double* heights, double* rMu,
                     double* rVar, double* rates, int64_t ld_state);
/* post: [batch][3] = ln prior, ln likelihood, ln jacobianRootBranch of the current states. */
int mcd_mh_get_posterior(const mcd_mh_t* m, double* post);
/*
 * n_iter iterations of steps_per_iter proposals; schedule[n_iter * steps_per_iter] = proposal row per step (host).
 * accumulate != 0: after every iteration add the absolute node ages tH * h_v to the running sums.
 * trace_alpha / trace_accept (host, may be NULL): [n_iter * steps_per_iter][batch] ln acceptance ratio / decision.
 * Trees of at most 64 nodes: the whole schedule in one launch, the factor of Sigma in LDS; up to 514 nodes (N <= 512) and 1024 chains: the
 * same with the factor streamed through LDS once per step (default up to 258 nodes).  Trees of 259 .. 1026 nodes: every run of steps between
 * two proposals that move more than 192 branch distances in one launch, the chains' states in LDS (k_mh_segment.hip); such a dense
 * proposal by two launches.  Trees of up to 258 nodes with more than 1024 chains, and trees over a sparse likelihood: two launches per step -- accept the pending
 * proposal and propose the next one; then ln likelihood of the proposed states, which up to 256 dimensions also carries their ln
 * prior as workgroups of a second role (both depend on the proposal only); from 321 nodes the likelihood launch only for proposals that
 * move more than 32 distances (the others: columns of L^-1 on a kept z).  Knobs (mcd_set_option), for tests and timing:
 * MCD_MH_SEGMENTS=0 / MCD_MH_INCREMENTAL=0 switch the segments / every incremental evaluation off (the chains: same decisions and states),
 * MCD_MH_INC_SLOTS (consulted by mcd_mh_create) sets the number of moved distances up to which a proposal counts as sparse (at most 256),
 * MCD_MH_SPARSE_SLOTS the same for the whole-schedule kernel of 65 .. 258 nodes (at most 64), MCD_MH_CHAIN_LW=0 runs the small-tree kernel with
 * one wave per chain (the likelihood then after the prior instead of beside it: the same bits),
 * MCD_MH_PRIOR=0 evaluates the prior inside the first launch everywhere (the chains are the same bits either way),
 * MCD_MH_PER_PHASE=1 takes the two-launch path for small trees as well, MCD_MH_STEP_WG=1 / 0 forces / forbids the step kernel
 * with a workgroup per chain (default: trees of more than 320 nodes).
 */
int mcd_mh_run(mcd_mh_t* m, const int32_t* schedule, int64_t n_iter, int32_t steps_per_iter, int accumulate,
               double* trace_alpha, int8_t* trace_accept);
/* Which launch structure the last mcd_mh_run took (reporting: bench.py, tests). */
#define MCD_MH_PATH_NONE 0
#define MCD_MH_PATH_CHAIN_LDS 1               /* whole schedule in one launch, factor resident in LDS (<= 64 nodes) */
#define MCD_MH_PATH_CHAIN_STREAMED 2          /* whole schedule in one launch, factor streamed once per step (k_mh_chain_big) */
#define MCD_MH_PATH_TWO_LAUNCH_PRIOR_BESIDE 3 /* step + likelihood launch that also carries the ln prior of the proposal */
#define MCD_MH_PATH_TWO_LAUNCH 4              /* step (prior inside) + likelihood launch */
#define MCD_MH_PATH_STEP_WG_X 5               /* workgroup-per-chain step leaving distances + plain-vector likelihood launch */
#define MCD_MH_PATH_STEP_WG_INCREMENTAL 6     /* the same, the likelihood launch only for proposals that move many distances (k_mh_inc.hip) */
#define MCD_MH_PATH_STEP_WG_SPARSE 7          /* workgroup-per-chain step leaving distances + the sparse product on them (mcd_mh_create_sparse) */
#define MCD_MH_PATH_SEGMENTS 8                /* 259 .. 1026 nodes: every run of steps between two dense proposals in one launch (state in LDS); a dense proposal: proposed by the segment before it (else by path 6's step kernel) + the row-split launch */
#define MCD_MH_PATH_SEGMENTS_SPARSE 9         /* sparse likelihood, 3 .. 2048 nodes: the same with the quadratic form updated through the rows of the moved distances (k_mh_segment_sparse.hip); a dense proposal: proposed by the segment before it + the one-launch full form */
int mcd_mh_last_path(const mcd_mh_t* m);
/* LDS bytes per workgroup of the persistent kernel the last mcd_mh_run launched (reporting: a profiler's kernel trace shows the static group
 * segment only, which is 0 for the kernels that size their LDS at launch); 0 when the run took per-step launches only. */
int64_t mcd_mh_last_dynamic_lds(const mcd_mh_t* m);
/* Auto tuning at the end of a tuning period [mcmc]: t' = clamp(t exp(2 (rate - optimal(dim))), 1e-5, 1e3). */
int mcd_mh_tune(mcd_mh_t* m);
/* tune: [batch][n_prop]; accepted / tried: counters since the last mcd_mh_tune / mcd_mh_reset_counters. */
int mcd_mh_get_tuning(const mcd_mh_t* m, double* tune, int32_t* accepted, int32_t* tried);
int mcd_mh_set_tuning(mcd_mh_t* m, const double* tune);
int mcd_mh_reset_counters(mcd_mh_t* m);
/* Reciprocal temperatures beta[batch] in (0, 1] (default 1): chain b accepts with (prior x likelihood)^beta[b], the
 * heated chains of Metropolis-coupled MCMC (`mc3`, app/Main.hs:476-478; package `mcmc`). */
int mcd_mh_set_temperatures(mcd_mh_t* m, const double* beta);
/*
 * The swap phase of Metropolis-coupled MCMC on the device.  Replaces: the swap bookkeeping of `mc3 (MC3Settings (NChains 4)
 * (SwapPeriod 2) (NSwaps 3))`, app/Main.hs:476-478 (package `mcmc`; its initial ladder and ladder tuning are not restated).
 *   mcd_mh_mc3_init   groups of n_chains consecutive GLOBAL chains (total_chains of them over all GPUs, a multiple of n_chains;
 *                     this handle holds [first_chain, first_chain + batch), mcd_mh_set_chain_offset), ladder betas[n_chains]
 *                     (betas[0] = 1, decreasing); chain c starts with rank c mod n_chains; sets this handle's temperatures.
 *   mcd_mh_mc3_swap   one phase: in every group n_swaps distinct adjacent rank pairs, in random order, exchange their
 *                     TEMPERATURES with probability min(1, exp((beta_i - beta_j)(ln pi_j - ln pi_i))), pi = prior x likelihood
 *                     (the states stay where they are: the same Markov chain, 8 bytes per chain on the wire).  gathered =
 *                     device pointer [world][3][chains_per_rank], what mcd_shard_allgather makes of the ranks'
 *                     mcd_mh_posterior_device arrays; NULL when the handle holds every chain.  Every rank evaluates all
 *                     groups from the gathered values with Philox numbers keyed (seed, group, phase): all ranks hold the same
 *                     table and the run does not depend on the number of GPUs.  Enqueued on the sampler's stream.
 *   mcd_mh_mc3_get    rank[total_chains], swap counters tried / accepted[n_chains - 1] per rung, beta[batch] (any may be NULL).
 * One period of the reference's loop = mcd_mh_run (SwapPeriod iterations) -> mcd_shard_allgather of the posterior on the
 * sampler's stream -> mcd_mh_mc3_swap; monitors read the chains whose rank is 0.
 */
int mcd_mh_mc3_init(mcd_mh_t* m, int n_chains, const double* betas, int64_t total_chains, uint64_t seed);
int mcd_mh_mc3_swap(mcd_mh_t* m, int n_swaps, const double* gathered, int world, int64_t chains_per_rank);
int mcd_mh_mc3_get(const mcd_mh_t* m, int32_t* rank, int64_t* tried, int64_t* accepted, double* beta);
/* age_sum / age_sq: [batch][n_nodes] running sums over *n_samples accumulated iterations; reset with the call below. */
int mcd_mh_get_age_sums(const mcd_mh_t* m, double* age_sum, double* age_sq, int64_t* n_samples);
int mcd_mh_reset_age_sums(mcd_mh_t* m);

/* ------------------------------------------------------------------------------------------------
 * The sparse form: the precision matrix as it is, in CSR on the device, no densification; N up to MCD_MAX_SPARSE_DIM.
 * Replaces: logDensitySparseMultivariateNormal (app/Probability.hs:178-184) and the closure likelihoodFunction (Sparse mu
 * sigmaInvSparse logDetSigma) (:279) over the operands of getData's SparseS branch (app/Main.hs:95-97: the association list
 * [((i, j), v)] of `prepare`'s graphical-lasso estimate, :142-155, 257-277) -- the reference's route for trees with thousands
 * of branches (tutorial/main/tutorial.org:487-496), beyond MCD_MAX_DIM of the dense kernels.
 *   ll = -N ln sqrt(2 pi) - 1/2 (logdet_sigma + dx^T P dx),  dx = x - mu;   mcd_sparse_grad_batch also returns its gradient
 *   G = -1/2 (P + P^T) dx  (= -P dx for the symmetric matrices `prepare` writes).
 * row / col / val: the nnz entries in any order, entries of one position are added up; P is used as given (no symmetrisation,
 * no positive-definiteness check: the reference evaluates the form with whatever the .data file holds).  Batches are chain-major
 * like everywhere; trees as in mcd_tree_create.  Up to 4096 chains the log-density is ONE launch (a workgroup stages the dx of one or two
 * chains in LDS and walks the flat stream of the matrix's entries; for an exactly symmetric matrix the upper triangle only), beyond that
 * three launches with lanes = chains; the two agree to rounding.  (mcd_mvn_create with a densified matrix remains the route to the gradient
 * of tree states and to the NUTS driver, N <= MCD_MAX_DIM; the Metropolis-Hastings driver takes the sparse handle: mcd_mh_create_sparse.)
 * mcd_sparse_release_stream: a host that makes short-lived streams calls it before destroying one (the gradient's and the large batches'
 * scratch buffer of that stream is freed; like mcd_mvn_release_stream).
 */
#define MCD_MAX_SPARSE_DIM 8192
typedef struct mcd_sparse mcd_sparse_t;
typedef struct mcd_sparse_tree mcd_sparse_tree_t;
int mcd_sparse_create(mcd_sparse_t** out, int n, const double* mu, int64_t nnz, const int32_t* row, const int32_t* col, const double* val,
                      double logdet_sigma, int device_id);
void mcd_sparse_destroy(mcd_sparse_t* h);
int mcd_sparse_dim(const mcd_sparse_t* h);
int64_t mcd_sparse_nnz(const mcd_sparse_t* h);   /* stored entries after adding up duplicates */
int mcd_sparse_release_stream(const mcd_sparse_t* h, void* stream /* hipStream_t */);
int mcd_sparse_logpdf_batch(const mcd_sparse_t* h, const double* X, int64_t ld, int64_t batch, int on_device, void* stream, double* ll);
int mcd_sparse_grad_batch(const mcd_sparse_t* h, const double* X, int64_t ld, int64_t batch, int on_device, void* stream, double* ll,
                          double* G, int64_t ldg);
/* State -> ln likelihood (likelihoodFunctionWrapper, app/Probability.hs:195-207) and ln jacobianRootBranch (:393-410) over a sparse
 * precision matrix; arguments as mcd_tree_loglik_batch. */
int mcd_sparse_tree_create(mcd_sparse_tree_t** out, const mcd_sparse_t* h, int n_nodes, const int32_t* parent);
void mcd_sparse_tree_destroy(mcd_sparse_tree_t* t);
int mcd_sparse_tree_loglik_batch(const mcd_sparse_tree_t* t, const double* heights, const double* rates, int64_t ld_state, const double* tH,
                                 const double* rMu, int64_t batch, int on_device, void* stream, double* ll, double* log_jac);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU.  Chains are independent: every rank (one process per GPU) holds the operands and evaluates its own
 * contiguous block of chains; nothing is exchanged on the likelihood path.  The sampler-level exchange -- the per-chain ln
 * posterior that MC3's swap phase compares (`mc3 (MC3Settings (NChains 4) (SwapPeriod 2) (NSwaps 3))`, app/Main.hs:476-478; in
 * the reference an in-process matter of `mcmc`), diagnostics -- is ONE all-gather per swap period:
 *   mcd_shard_allgather   recv[r * count + i] = send of rank r, element i; device pointers, enqueued on `stream`
 *                         (RCCL ncclAllGather over xGMI; a few KB per rank, latency bound -- no ring of small sends);
 *   mcd_mh_posterior_device  the sampler's device-resident [3][batch] ln prior / ln likelihood / ln Jacobian, what to send.
 * RCCL is loaded at run time; the communicator is made by the wrappers below, so that a host in the reference's language
 * needs no RCCL binding: rank 0 draws an id (mcd_shard_unique_id), hands its MCD_SHARD_ID_BYTES bytes to the other ranks'
 * processes by its own means, every rank calls mcd_shard_comm_create (collective: returns when all ranks have called).
 */
#define MCD_SHARD_ID_BYTES 128
int mcd_shard_unique_id(char id[MCD_SHARD_ID_BYTES]);
int mcd_shard_comm_create(void** comm, int world_size, int rank, const char id[MCD_SHARD_ID_BYTES], int device_id);
void mcd_shard_comm_destroy(void* comm);
int mcd_shard_comm_count(void* comm, int* n_ranks);   /* the rank count the RCCL communicator itself reports (ncclCommCount) */
int mcd_shard_allgather(void* comm, const double* send, double* recv, int64_t count, void* stream);
int mcd_mh_posterior_device(const mcd_mh_t* m, const double** post, void** stream);

#ifdef __cplusplus
}
#endif
#endif /* MCMCDATE_MVN_H */
