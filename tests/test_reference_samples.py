"""CPU tests: the oracle's Metropolis-Hastings twin (oracle/mh_oracle.c + mvn_oracle.c + prior_oracle.c) against outputs the
reference itself commits -- the node ages sampled by its six posterior and six prior-only chains on the 7-taxon mtCDNApri
analysis (bench/comparison_with_mcmctree/03_compare_estimates/{post,prior}_samples_run{1..6}.tsv, condensed by
tests/golden/make_{post,prior}_sample_summary.py).  These pin the ORACLE end to end (prepare with the graphical lasso,
likelihood, prior, the whole proposal cycle with its Jacobians and root-branch lifts); the `-m gpu` tests compare the HIP path
with this twin step by step and with the same fixtures at the sampler level (tests/test_gpu_mh.py).

Round-2 review, item 1: the prior-only ROOT age was 22.4 against the reference's 19.0.  Found in round 3: not a semantic of
`liftProposalWith` / `scaleUnbiased` / `scaleContrarily` but the input -- the committed samples carry a soft upper bound of the
root at 30.0 (tail mass 0.025), the committed calibration tree says 'U(100,2.5e-2)'.  test_the_samples_carry_a_root_bound_of_30
is the single experiment that shows it; with that bound every one of the six ages, means and quantiles, is the reference's."""
import numpy as np
import pytest

import mtcdnapri as A

CHAINS = 16          # 16 chains x 12 930 iterations x 251 proposals: about 20 s on 8 cores


def summary(ages, nodes):
    s = ages[:, :, nodes].reshape(-1, len(nodes))
    q = np.quantile(s, [0.025, 0.975], axis=0)
    return s, s.mean(axis=0), s.std(axis=0, ddof=1), q[0], q[1]


@pytest.fixture(scope="module")
def prior_runs():
    """The prior-only analysis on the twin with the committed root bound (100) and with the bound the samples carry (30)."""
    out = {}
    for upper in (None, A.ROOT_UPPER_OF_THE_SAMPLES):
        an = A.analysis("NoLikelihood", root_upper=upper)
        out[upper] = A.monitored_ages(A.twin_chains(an, CHAINS, seed=7), an.table, seed=7)
    return out


def test_twin_reproduces_the_references_posterior_samples():
    """`./run -s -f analysis.conf -c ul s r` on the CPU twin: all six node ages within 1 % of the reference's pooled means
    (north_star's bar; measured 0.1 .. 0.5 %), the 2.5 / 97.5 % quantiles within 3 %, with the calibration tree as committed."""
    post = A.golden("post")
    ref = {k: np.array(v) for k, v in post["pooled"].items()}
    an = A.analysis("SparseMultivariateNormal 0.1")
    assert an.ht == 50.0                                                   # getMeanRootHeight of U(100, .)
    ages = A.monitored_ages(A.twin_chains(an, CHAINS, seed=3), an.table, seed=3)
    _, mean, sd, q025, q975 = summary(ages, post["nodes"])
    assert np.all(np.abs(mean - ref["mean"]) <= 0.01 * ref["mean"]), (mean, ref["mean"])
    assert np.all(np.abs(q025 - ref["q025"]) <= 0.03 * ref["q025"]), (q025, ref["q025"])
    assert np.all(np.abs(q975 - ref["q975"]) <= 0.03 * ref["q975"]), (q975, ref["q975"])
    assert np.all(np.abs(sd - ref["sd"]) <= 0.06 * ref["sd"]), (sd, ref["sd"])


def test_twin_reproduces_the_references_prior_only_samples(prior_runs):
    """`./run -s -f analysis.conf -c ul n r` on the CPU twin with the root bound of the samples: ALL SIX node ages, the root
    included, within 1 % of the reference's pooled means (measured <= 0.6 %), standard deviations within 3 %, the 2.5 / 97.5 %
    quantiles within 2 % (nodes 5 and 9 reach down to 0.1: their lower quantile is compared on the scale of the mean), and the
    joint structure -- correlations between the ages within 0.03, mean relative heights node / root within 1 %."""
    fx = A.golden("prior")
    ref = {k: np.array(v) for k, v in fx["pooled"].items()}
    s, mean, sd, q025, q975 = summary(prior_runs[A.ROOT_UPPER_OF_THE_SAMPLES], fx["nodes"])
    assert np.all(np.abs(mean - ref["mean"]) <= 0.01 * ref["mean"]), (mean, ref["mean"])
    assert np.all(np.abs(sd - ref["sd"]) <= 0.03 * ref["sd"]), (sd, ref["sd"])
    assert np.all(np.abs(q025 - ref["q025"]) <= 0.02 * ref["mean"]), (q025, ref["q025"])
    assert np.all(np.abs(q975 - ref["q975"]) <= 0.02 * ref["q975"]), (q975, ref["q975"])
    assert np.max(np.abs(np.corrcoef(s.T) - np.array(fx["correlation"]))) <= 0.03
    rel = (s[:, 1:] / s[:, :1]).mean(axis=0)
    assert np.all(np.abs(rel - np.array(fx["relative_height"]["mean"])) <= 0.01 * np.array(fx["relative_height"]["mean"]))


@pytest.mark.xfail(strict=True, reason="REPORTED DISCREPANCY, kept visible: with the calibration file AS COMMITTED ('U(100,2.5e-2)' on the root) the prior-only "
                                        "root age is 15 % above the reference's committed samples; the match above holds under the hypothesis -- inferred from "
                                        "the samples' own edge, not verifiable from the reference's sources -- that they were drawn with a root bound of 30")
def test_prior_only_root_age_with_the_calibration_file_as_committed(prior_runs):
    """The same comparison with the reference's committed input unchanged.  Expected to fail (strict): the day this passes, the inferred
    bound is no longer needed and the hypothesis should be dropped from README / DESIGN."""
    fx = A.golden("prior")
    ref = {k: np.array(v) for k, v in fx["pooled"].items()}
    _, mean, _, _, _ = summary(prior_runs[None], fx["nodes"])
    assert np.all(np.abs(mean - ref["mean"]) <= 0.01 * ref["mean"]), (mean, ref["mean"])


def test_the_samples_carry_a_root_bound_of_30(prior_runs):
    """The experiment that localises round 2's mismatch.  (1) The reference's root ages stop: no value above 31.6 in 29 100
    samples, the largest of every run within 31.3 .. 31.6, while under the committed bound 'U(100,2.5e-2)' 10 % of the twin's
    root ages lie above 32.  (2) Below the edge the two distributions are the same: the twin's samples under the committed bound,
    cut at 30.4, have the reference's mean and spread on every node.  (3) The edge has the shape of calibrateSoftF's
    half-normal tail (Calibration.hs:369-391: standard deviation sqrt(2/pi) x 0.025 on the relative scale): a maximum-likelihood
    fit of (bound, width) on the reference's root ages above 26, with the twin's power-law decay as base density, gives
    bound 30.0 +- 0.15 and the theoretical width within 15 %."""
    fx = A.golden("prior")
    ref = {k: np.array(v) for k, v in fx["pooled"].items()}
    run_max = np.array(fx["root_age_max_per_run"])
    assert run_max.max() < 31.6 and run_max.min() > 31.2
    s100 = summary(prior_runs[None], fx["nodes"])[0]
    assert (s100[:, 0] > 32.0).mean() > 0.10 and s100[:, 0].mean() > 1.15 * ref["mean"][0]      # round 2's 22.4 against 19.0
    cut = s100[s100[:, 0] < 30.4]
    assert np.all(np.abs(cut.mean(axis=0) - ref["mean"]) <= 0.01 * ref["mean"]), (cut.mean(axis=0), ref["mean"])
    assert np.all(np.abs(cut.std(axis=0) - ref["sd"]) <= 0.03 * ref["sd"])
    # edge fit: density on [26, 34] proportional to t^-k w(t; b, s), w = 1 below b, exp(-(1 - b/t)^2 / (2 s^2)) above
    from scipy.optimize import minimize

    lo, hi = 26.0, 34.0
    o = s100[(s100[:, 0] > lo) & (s100[:, 0] < hi), 0]
    k = minimize(lambda k: (k[0] * np.log(o)).sum() + len(o) * np.log((lo ** (1 - k[0]) - hi ** (1 - k[0])) / (k[0] - 1)), [4.0],
                 method="Nelder-Mead").x[0]
    r = np.array(fx["root_ages_above_26"])
    tt = np.linspace(lo, hi, 8001)

    def w(t, b, s):
        ex = np.maximum(1.0 - b / t, 0.0)
        return np.exp(-ex * ex / (2 * s * s))

    def nll(p):
        dens = tt ** -k * w(tt, p[0], p[1])
        z = np.sum(0.5 * (dens[1:] + dens[:-1])) * (tt[1] - tt[0])
        return -np.log(r ** -k * w(r, p[0], p[1])).sum() + len(r) * np.log(z)

    b, sw = minimize(nll, [29.0, 0.03], method="Nelder-Mead").x
    assert abs(b - A.ROOT_UPPER_OF_THE_SAMPLES) <= 0.15, (b, sw)
    assert abs(sw - 0.7978845608028654 * 0.025) <= 0.15 * 0.7978845608028654 * 0.025, (b, sw)
    # the posterior runs stop at the same place (their root age has mean 17.2, sd 2.2: 30.9 is six sd out -- an edge, not a tail)
    assert max(A.golden("post")["root_age_max_per_run"]) < 31.0
