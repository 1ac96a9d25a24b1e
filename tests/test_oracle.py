"""CPU tests of the oracle (oracle/mvn_oracle.c) against the committed golden fixtures, an independent
numpy/scipy implementation, and the reference's documented conventions.

The reference ships no expected outputs for this path (parity unpinned, see oracle/mvn_oracle.c);
these tests pin the restatement against (a) the fixtures generated from the reference's own test
inputs, (b) scipy.stats.multivariate_normal, (c) the worked micro-example of the branch order.
"""
import numpy as np
import pytest
from scipy.stats import multivariate_normal

import oracle as O
from oracle import prepare as P

FIX = ["06-leaves-constant-rate", "10-leaves-autocorrelated-rate", "12-leaves-variable-rate", "24-leaves-braces",
       "25-leaves-bastien"]


def rel(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


@pytest.mark.parametrize("name", FIX)
def test_oracle_matches_golden_loglik(golden, name):
    fx = golden[name]
    ll = O.logpdf_full_batch(fx["mu"], fx["sigma_inv"], float(fx["logdet"]), fx["X"])
    assert np.array_equal(ll, fx["ll_X"])          # same code, same machine arithmetic: bit-exact
    ll_s, lj_s = O.tree_loglik_full_batch(fx["parent"], fx["H"], fx["R"], fx["tH"], fx["rMu"], fx["mu"],
                                          fx["sigma_inv"], float(fx["logdet"]))
    assert np.array_equal(ll_s, fx["ll_S"]) and np.array_equal(lj_s, fx["lj_S"])


@pytest.mark.parametrize("name", FIX)
def test_oracle_against_scipy_and_cholesky_form(golden, name):
    fx = golden[name]
    ll = O.logpdf_full_batch(fx["mu"], fx["sigma_inv"], float(fx["logdet"]), fx["X"])
    ref = multivariate_normal(mean=fx["mu"], cov=fx["sigma"]).logpdf(fx["X"])
    assert np.max(rel(ll, ref)) <= 1e-9
    L = O.cholesky(fx["sigma"])
    assert np.allclose(L @ L.T, fx["sigma"], rtol=1e-13, atol=1e-14 * np.abs(fx["sigma"]).max())
    ll_chol = O.logpdf_chol_batch(fx["mu"], L, fx["X"])
    assert np.max(rel(ll, ll_chol)) <= 1e-10       # Sigma^-1 form (reference) vs Cholesky form (north_star)
    # long double arbiter
    for i in (0, 17, len(ll) - 1):
        assert abs(O.logpdf_full_ld(fx["mu"], fx["sigma_inv"], float(fx["logdet"]), fx["X"][i]) - ll[i]) <= 1e-10 * max(1, abs(ll[i]))
    # documented values of SURVEY.md 8c
    expect = {"06-leaves-constant-rate": (9, -76.117828, 29.788467), "12-leaves-variable-rate": (21, -189.998997, 75.701790),
              "24-leaves-braces": (45, -405.463076, 161.379304)}
    if name in expect:
        k, logdet, ll_mu = expect[name]
        assert len(fx["mu"]) == k and abs(float(fx["logdet"]) - logdet) < 1e-5 and abs(ll[-1] - ll_mu) < 1e-5


@pytest.mark.parametrize("name", FIX)
def test_tree_wrapper_against_numpy_twin(golden, name):
    fx = golden[name]
    for b in range(0, len(fx["tH"]), 5):
        d_c = O.distances(fx["parent"], fx["H"][b], fx["R"][b], fx["tH"][b], fx["rMu"][b])
        d_np = P.distances_np(fx["parent"], fx["H"][b], fx["R"][b], fx["tH"][b], fx["rMu"][b])
        assert np.allclose(d_c, d_np, rtol=1e-15, atol=0)
        ll = P.logpdf_full_np(fx["mu"], fx["sigma_inv"], float(fx["logdet"]), d_np)
        assert abs(ll - fx["ll_S"][b]) <= 1e-9 * max(1.0, abs(ll))
        # jacobianRootBranch = log (1 / d_0), app/Probability.hs:393-410
        assert abs(fx["lj_S"][b] - np.log(1.0 / d_c[0])) <= 1e-14


def test_gradient_against_finite_differences(golden):
    fx = golden["24-leaves-braces"]
    b = 2
    gH, gR, gt, gm = O.tree_grad_full(fx["parent"], fx["H"][b], fx["R"][b], fx["tH"][b], fx["rMu"][b], fx["mu"], fx["sigma_inv"])
    assert np.array_equal(gH, fx["gH"][b]) and np.array_equal(gR, fx["gR"][b])

    def f(H, R, tH, rMu):
        return O.tree_loglik_full_batch(fx["parent"], H[None], R[None], np.array([tH]), np.array([rMu]), fx["mu"],
                                        fx["sigma_inv"], float(fx["logdet"]))[0][0]

    H0, R0, t0, m0 = fx["H"][b], fx["R"][b], fx["tH"][b], fx["rMu"][b]
    for v in range(1, len(H0)):
        for which, g in (("H", gH), ("R", gR)):
            x = (H0 if which == "H" else R0)
            if x[v] == 0.0:
                continue
            h = 1e-6 * abs(x[v])
            xp, xm = x.copy(), x.copy()
            xp[v] += h
            xm[v] -= h
            fd = (f(xp, R0, t0, m0) - f(xm, R0, t0, m0)) / (2 * h) if which == "H" else (f(H0, xp, t0, m0) - f(H0, xm, t0, m0)) / (2 * h)
            assert abs(fd - g[v]) <= 2e-5 * max(1.0, abs(fd)), (which, v, fd, g[v])
    h = 1e-6
    assert abs((f(H0, R0, t0 * (1 + h), m0) - f(H0, R0, t0 * (1 - h), m0)) / (2 * h * t0) - gt) <= 1e-5 * abs(gt)
    assert abs((f(H0, R0, t0, m0 * (1 + h)) - f(H0, R0, t0, m0 * (1 - h))) / (2 * h * m0) - gm) <= 1e-5 * abs(gm)
    # raw-x gradient = -Sigma^-1 (x - mu)
    G = O.grad_full_batch(fx["mu"], fx["sigma_inv"], fx["X"][:4])
    assert np.allclose(G, -(fx["X"][:4] - fx["mu"]) @ fx["sigma_inv"], rtol=1e-12, atol=1e-9)


def test_branch_order_micro_example():
    """SURVEY.md Appendix A worked example (L = 3): root 1.0 with children x (0.4; leaves a, b) and leaf c."""
    parent = np.array([-1, 0, 1, 1, 0], np.int32)          # root, x, a, b, c in pre-order
    heights = np.array([1.0, 0.4, 0.0, 0.0, 0.0])
    t = O.height_to_length(parent, heights)                  # lib/Mcmc/Tree/Types.hs:224-233
    assert np.array_equal(t, [0.0, 0.6, 0.4, 0.4, 1.0])
    assert np.array_equal(O.get_branches(parent, t), [0.6, 1.0, 0.4, 0.4])          # app/Tools.hs:36-43
    rates = np.array([0.0, 2.0, 3.0, 5.0, 7.0])              # stem, r_x, r_a, r_b, r_c
    assert np.array_equal(O.get_branches(parent, rates), [2.0, 7.0, 3.0, 5.0])
    assert np.array_equal(O.sum_first_two(np.array([1.0, 2.0, 3.0, 4.0])), [3.0, 3.0, 4.0])   # app/Tools.hs:47-48
    s = 1.5 * 0.5
    d = O.distances(parent, heights, rates, 1.5, 0.5)
    assert np.allclose(d, s * np.array([0.6 * 2.0 + 1.0 * 7.0, 0.4 * 3.0, 0.4 * 5.0]), rtol=1e-15)
    # gradient back-propagation of the same example: d ll/d r_x = s g0 0.6, d ll/d h_x = s (g1 r_a + g2 r_b - g0 r_x)
    mu = np.array([5.0, 1.0, 1.5])
    Pm = np.array([[2.0, 0.3, 0.1], [0.3, 1.5, 0.2], [0.1, 0.2, 1.0]])
    gH, gR, gt, gm = O.tree_grad_full(parent, heights, rates, 1.5, 0.5, mu, Pm)
    g = -Pm @ (d - mu)
    assert np.allclose(gR[1:], [s * g[0] * 0.6, s * g[1] * 0.4, s * g[2] * 0.4, s * g[0] * 1.0], rtol=1e-13)
    assert abs(gH[1] - s * (g[1] * 3.0 + g[2] * 5.0 - g[0] * 2.0)) <= 1e-13 * abs(gH[1])
    assert abs(gt - (g @ d) / 1.5) <= 1e-13 * abs(gt) and abs(gm - (g @ d) / 0.5) <= 1e-13 * abs(gm)


def test_structural_errors():
    # root with three children: "getBranches: Root node is not bifurcating." (app/Tools.hs:43)
    with pytest.raises(O.OracleError, match="not bifurcating"):
        O.get_branches(np.array([-1, 0, 0, 0], np.int32), np.zeros(4))
    with pytest.raises(O.OracleError):
        O.cholesky(np.array([[1.0, 2.0], [2.0, 1.0]]))


def test_univariate_and_sparse_variants(golden):
    fx = golden["12-leaves-variable-rate"]
    mu, x = fx["mu"], fx["X"][3]
    vs = np.diag(fx["sigma"])
    ref = np.sum(-0.5 * np.log(2 * np.pi * vs) - 0.5 * (x - mu) ** 2 / vs)
    assert abs(O.logpdf_univariate(mu, vs, x) - ref) <= 1e-10 * abs(ref)
    Pm = fx["sigma_inv"]
    ii, jj = np.nonzero(np.ones_like(Pm))
    assert abs(O.logpdf_sparse(mu, ii, jj, Pm[ii, jj], float(fx["logdet"]), x)
               - O.logpdf_full(mu, Pm, float(fx["logdet"]), x)) <= 1e-10 * abs(ref)


def test_nonfinite_propagation(golden):
    fx = golden["06-leaves-constant-rate"]
    x = fx["X"][0].copy()
    x[2] = np.nan
    assert np.isnan(O.logpdf_full(fx["mu"], fx["sigma_inv"], float(fx["logdet"]), x))
    x[2] = np.inf
    assert not np.isfinite(O.logpdf_full(fx["mu"], fx["sigma_inv"], float(fx["logdet"]), x))
