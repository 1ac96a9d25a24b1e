"""CPU tests of the prior restatement (oracle/prior_oracle.c).

The birth-death part is PINNED by the reference's own known-answer values (comments in
lib/Mcmc/Tree/Prior/BirthDeath.hs:51-52 and :252-271, the latter cross-checked against RevBayes by the
reference's author).  The elementary densities come from the third-party `mcmc` package and are checked
against scipy.stats (independent implementation of the same textbook definitions)."""
import numpy as np
import pytest
from scipy import stats

import oracle as O
from oracle import prepare as P

L = O.lib


def test_compute_de_known_answer():
    # BirthDeath.hs:51-52  ">>> computeDE 1.2 3.2 1.0 0.3" (signature of that time: la mu dt e0, rho = 1)
    d, e = O.compute_de(1.2, 3.2, 1.0, 1.0, 0.3)
    assert d == pytest.approx(7.283127121752474e-2, rel=1e-15) and e == pytest.approx(0.9305035687810801, rel=1e-15)
    # near-critical variant is the la -> mu limit of the same formulas
    d0, e0 = O.compute_de(2.0, 2.0 - 1e-7, 0.7, 0.4, 0.2)
    d1, e1 = O.compute_de(2.0, 2.0 - 1e-7, 0.7, 0.4, 0.2, near_critical=True)
    assert abs(d0 - d1) < 1e-6 and abs(e0 - e1) < 1e-6


def test_birth_death_known_answers():
    # BirthDeath.hs:252-254: single leaf, stem 1, WithStem (= ConditionOnTimeOfOrigin)
    assert np.exp(O.birth_death(False, 1.2, 3.2, 1.0, [-1], [1.0])) == pytest.approx(5.8669248906043234e-2, rel=1e-14)
    # :256-258: that value predates the `br <= 0 = (0.0, 1.0)` guard (:200); with a zero stem the root term was
    # la * dL * dR, i.e. la times today's ConditionOnTimeOfMrca value
    v = O.birth_death(True, 1.2, 3.2, 1.0, [-1, 0, 0, 2, 2], [0.0, 0.4, 0.2, 0.2, 0.2])
    assert 1.2 * np.exp(v) == pytest.approx(4.3357752474276125e-2, rel=1e-14)
    assert O.birth_death(False, 1.2, 3.2, 1.0, [-1, 0, 0, 2, 2], [0.0, 0.4, 0.2, 0.2, 0.2]) == -np.inf   # today's guard
    # :260-271, "checked against RevBayes"
    t = P.parse_newick("(((a:1.0,b:1.0):1.0,c:2.0):1.0,d:3.0):0.0;")
    for mu, e in zip([0, 0.01, 0.05, 0.1, 0.2, 0.5],
                     [-10.09861228866811, -10.07675364864067, -9.993307032921498, -9.898174270006024, -9.73975910235509, -9.54137886890279]):
        assert np.log(1 / 3) + O.birth_death(True, 1.0, mu, 1.0, t.parent, t.length) == pytest.approx(e, rel=1e-14)
    for rho, e in zip([1, 0.9, 0.8], [-10.09861228866811, -9.809211822253452, -9.498032504556043]):
        assert np.log(1 / 3) + O.birth_death(True, 1.0, 0.0, rho, t.parent, t.length) == pytest.approx(e, rel=1e-14)
    assert np.log(1 / 3) + O.birth_death(True, 0.2, 0.5, 0.8, t.parent, t.length) == pytest.approx(-9.700151607658995, rel=1e-14)
    # structural faults (`error` in the reference) -> NaN
    assert np.isnan(O.birth_death(True, -1.0, 0.5, 1.0, t.parent, t.length))
    assert np.isnan(O.birth_death(True, 1.0, 0.5, 1.0, [-1, 0, 0, 0], [0, 1, 1, 1.0]))


def test_elementary_densities_against_scipy():
    lib = L()
    for x in (0.0, 0.3, 2.5):
        assert lib.orp_ln_exponential(1.7, x) == pytest.approx(stats.expon(scale=1 / 1.7).logpdf(x), rel=1e-13)
    assert lib.orp_ln_exponential(1.7, -0.1) == -np.inf
    for k, th, x in [(1.5, 1 / 6, 0.2), (4.0, 0.25, 1.3), (0.7, 2.0, 0.01)]:
        assert lib.orp_ln_gamma(k, th, x) == pytest.approx(stats.gamma(k, scale=th).logpdf(x), rel=1e-12)
    assert lib.orp_ln_gamma(1.5, 1 / 6, 0.0) == -np.inf
    assert lib.orp_ln_normal(0.3, 0.02, 0.31) == pytest.approx(stats.norm(0.3, 0.02).logpdf(0.31), rel=1e-13)
    # logNormal' m v x: log-normal with E[x] = m and log-variance v  (Yang 2006, eq. 7.23)
    m, v, x = 1.0, 0.4, 1.7
    ref = stats.lognorm(s=np.sqrt(v), scale=m * np.exp(-0.5 * v)).logpdf(x)
    assert lib.orp_ln_lognormal_prime(m, v, x) == pytest.approx(ref, rel=1e-12)
    assert lib.orp_ln_lognormal_prime(m, v, 0.0) == -np.inf


def test_soft_node_priors():
    lib = L()
    s = 0.7978845608028654 * 0.025
    # inside the interval: 1; outside: one-sided normal, continuous at the boundary (Calibration.hs:369-391)
    assert lib.orp_calibrate_soft(1, 0.4, 0.025, 1, 0.6, 0.025, 0.5) == 0.0
    assert lib.orp_calibrate_soft(1, 0.4, 0.025, 1, 0.6, 0.025, 0.39) == pytest.approx(-0.5 * (0.01 / s) ** 2, rel=1e-10)
    assert lib.orp_calibrate_soft(1, 0.4, 0.025, 1, 0.6, 0.025, 0.63) == pytest.approx(-0.5 * (0.03 / s) ** 2, rel=1e-10)
    assert lib.orp_calibrate_soft(0, 0.0, 0.0, 1, 0.6, 0.025, 0.01) == 0.0            # Zero lower boundary
    assert lib.orp_calibrate_soft(1, 0.4, 0.025, 0, 0.0, 0.0, 5.0) == 0.0             # Infinity upper boundary
    assert lib.orp_calibrate_soft(1, 0.4, 0.025, 1, 0.6, 0.025, -0.1) == -np.inf      # h < 0
    # constraint (Constraint.hs:403-415): younger below older -> 1
    assert lib.orp_constrain_soft(0.025, 0.3, 0.5) == 0.0
    assert lib.orp_constrain_soft(0.025, 0.52, 0.5) == pytest.approx(-0.5 * (0.02 / s) ** 2, rel=1e-10)
    # brace (Brace.hs:218-230)
    hs = np.array([0.5, 0.5, 0.5])
    assert lib.orp_brace_soft(1e-4, 3, hs.ctypes.data_as(O._dp)) == 0.0
    hs = np.array([0.5, 0.5002])
    assert lib.orp_brace_soft(1e-4, 2, hs.ctypes.data_as(O._dp)) == pytest.approx(2 * (-0.5 * (1e-4 / 1e-4) ** 2), rel=1e-9)


@pytest.mark.parametrize("model", ["UncorrelatedGamma", "UncorrelatedLogNormal", "UncorrelatedWhiteNoise", "AutocorrelatedLogNormal"])
def test_relaxed_clock_models_against_scipy(model):
    rng = np.random.default_rng(3)
    n = 9
    tlen = np.concatenate([[0.0], rng.uniform(0.05, 0.6, n - 1)])
    rates = np.concatenate([[0.0], np.exp(0.3 * rng.standard_normal(n - 1))])
    v = 0.35
    got = O.relaxed_clock(model, 1.0, v, tlen, rates)
    if model == "UncorrelatedGamma":
        ref = stats.gamma(1.0 / v, scale=v).logpdf(rates[1:]).sum()
    elif model == "UncorrelatedLogNormal":
        ref = stats.lognorm(s=np.sqrt(v), scale=np.exp(-0.5 * v)).logpdf(rates[1:]).sum()
    elif model == "UncorrelatedWhiteNoise":
        vv = v / tlen[1:]
        ref = sum(stats.gamma(1.0 / a, scale=a).logpdf(r) for a, r in zip(vv, rates[1:]))
    else:
        vv = v * tlen[1:]
        ref = sum(stats.lognorm(s=np.sqrt(a), scale=np.exp(-0.5 * a)).logpdf(r) for a, r in zip(vv, rates[1:]))
    assert got == pytest.approx(ref, rel=1e-12)


def test_full_prior_composition(golden):
    """priorFunction = node priors * birth-death block * relaxed-clock block (app/Probability.hs:127-150)."""
    fx = golden["12-leaves-variable-rate"]
    parent = fx["parent"]
    spec = O.PriorSpec(parent, ht=1050.0, model="UncorrelatedLogNormal",
                       calibrations=[(0, 900.0, 0.025, 1200.0, 0.025), (5, 400.0, 0.025, None, 0.0)],
                       constraints=[(3, 5, 0.025)], braces=[([3, 7], 1e-3)])
    b = 4
    H, R = fx["H"][b], fx["R"][b]
    total, comp = O.prior(spec, 0.8, 1.3, 1000.0, H, 7e-4, 0.4, R)
    tlen = O.height_to_length(parent, H)
    c1 = -0.8 - 1.3 + O.birth_death(True, 0.8, 1.3, 1.0, parent, tlen)
    c2 = np.log(1050.0) - 1050.0 * 7e-4 + stats.gamma(1.5, scale=1 / 6).logpdf(0.4) + O.relaxed_clock("UncorrelatedLogNormal", 1.0, 0.4, tlen, R)
    assert comp[1] == pytest.approx(c1, rel=1e-13) and comp[2] == pytest.approx(c2, rel=1e-13)
    assert total == pytest.approx(comp.sum(), rel=1e-15) and np.isfinite(total)
    # calibration of the root: tH * 1.0 inside [900, 1200] -> no penalty from it; h <= 0 -> probability 0
    assert O.prior(spec, 0.8, 1.3, -1.0, H, 7e-4, 0.4, R)[0] == -np.inf
