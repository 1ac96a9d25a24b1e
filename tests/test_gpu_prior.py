"""GPU parity of the batched prior kernel against the prior oracle and the golden
fixtures.  Tolerance: |lp_gpu - lp_oracle| <= 1e-11 * max(1, |lp|) per block and in total (fp64; the
kernel sums the per-node terms in a different order than the reference's recursion)."""
import numpy as np
import pytest

import mcmc_date_amd as M
import oracle as O

pytestmark = pytest.mark.gpu
MODELS = ["UncorrelatedGamma", "UncorrelatedLogNormal", "UncorrelatedWhiteNoise", "AutocorrelatedLogNormal"]
FIX = ["06-leaves-constant-rate", "10-leaves-autocorrelated-rate", "12-leaves-variable-rate", "24-leaves-braces", "25-leaves-bastien"]


def tables(fx):
    cal = [M.Calibration(f"c{i}", int(r[0]), r[2] if r[1] else None, r[3], r[5] if r[4] else None, r[6]) for i, r in enumerate(fx["cal"])]
    con = [M.Constraint(f"k{i}", int(r[0]), int(r[1]), r[2]) for i, r in enumerate(fx["con"])]
    br = [M.Brace(f"b{i}", [int(n) for n in fx["brace_nodes"][fx["brace_ptr"][i]:fx["brace_ptr"][i + 1]]], float(s))
          for i, s in enumerate(fx["brace_sd"])]
    return cal, con, br


def batch(fx):
    return M.StateBatch(fx["H"], fx["R"], fx["prior_tH"], fx["rMu"], fx["prior_birth"], fx["prior_death"], fx["prior_rvar"])


def close(a, b, rtol=1e-11):
    fin = np.isfinite(b)
    return np.array_equal(np.isfinite(a), fin) and np.all(np.abs(a[fin] - b[fin]) <= rtol * np.maximum(1.0, np.abs(b[fin]))) \
        and np.array_equal(a[~fin], b[~fin], equal_nan=True)


@pytest.mark.parametrize("name", FIX)
@pytest.mark.parametrize("model", MODELS)
def test_prior_matches_fixture(gpu, golden, name, model):
    fx = golden[name]
    topo = M.Topology(fx["parent"])
    cal, con, br = tables(fx)
    pf = M.PriorFunction(float(fx["prior_ht"]), model, cal, con, br, topo)
    s = batch(fx)
    lp, comp = pf.logprior(s, want_components=True)
    assert close(comp, fx["lpc_" + model]) and close(lp, fx["lp_" + model])
    lp_d = pf.logprior(s.to(gpu))
    assert np.array_equal(lp_d.cpu().numpy(), lp)
    # the plugin closure
    f = M.prior_function(float(fx["prior_ht"]), model, cal, con, br, topo)
    b = 2
    x = M.State(fx["prior_birth"][b], fx["prior_death"][b], fx["prior_tH"][b], fx["H"][b], fx["rMu"][b], fx["prior_rvar"][b], fx["R"][b])
    assert f(x) == lp[b]


def test_prior_edge_cases(gpu, golden):
    fx = golden["12-leaves-variable-rate"]
    topo = M.Topology(fx["parent"])
    cal, con, br = tables(fx)
    pf = M.PriorFunction(float(fx["prior_ht"]), "UncorrelatedGamma", cal, con, br, topo)
    spec = O.PriorSpec(fx["parent"], float(fx["prior_ht"]), "UncorrelatedGamma",
                       [(c.node, c.lower, c.lower_p, c.upper, c.upper_p) for c in cal], [(k.young, k.old, k.p) for k in con], [])
    s = batch(fx).slice(0, 8)
    H, R = s.heights.copy(), s.rates.copy()
    tH, birth, death, rvar, rmu = (a.copy() for a in (s.time_height, s.time_birth_rate, s.time_death_rate, s.rate_variance, s.rate_mean))
    H[1, 5] = H[1, fx["parent"][5]] + 0.01          # a negative branch: probability 0 from the birth-death prior
    R[2, 3] = -0.5                                  # a negative rate: probability 0 from the clock prior
    tH[3] = -1.0                                    # non-positive height multiplier: probability 0 (Combined.hs:78)
    rvar[4] = 0.0                                   # the reference calls `error`: NaN here
    birth[5] = -0.1                                 # `error` in birthDeath: NaN here
    rmu[6] = -1e-3                                  # exponential prior: probability 0
    lp = pf.logprior(M.StateBatch(H, R, tH, rmu, birth, death, rvar))
    ref = np.array([O.prior(spec, birth[b], death[b], tH[b], H[b], rmu[b], rvar[b], R[b])[0] for b in range(8)])
    assert np.isfinite(lp[0]) and np.isfinite(lp[7]) and close(lp[[0, 7]], ref[[0, 7]])
    assert lp[1] == -np.inf and lp[2] == -np.inf and lp[3] == -np.inf and lp[6] == -np.inf
    assert np.isnan(lp[4]) and np.isnan(lp[5])
    assert ref[1] == -np.inf and ref[2] == -np.inf and ref[3] == -np.inf and ref[6] == -np.inf and np.isnan(ref[4]) and np.isnan(ref[5])
    # structural faults at creation
    with pytest.raises(M.RootNotBifurcating):
        M.PriorFunction(1.0, "UncorrelatedGamma", [], [], [], M.Topology(np.array([-1, 0, 0, 0], np.int32)))
    with pytest.raises(M.McdError):
        M.PriorFunction(1.0, "UncorrelatedGamma", [M.Calibration("x", 99, 0.1, 0.025, None, 0.0)], [], [], topo)
    with pytest.raises(M.McdError):
        M.PriorFunction(1.0, "UncorrelatedGamma", [], [], [M.Brace("b", [1, 2], 0.0)], topo)
    with pytest.raises(ValueError):
        pf.logprior(M.StateBatch(H, R, tH, rmu))


def test_prior_large_tree(gpu):
    """More nodes than lanes: 129 leaves (257 nodes), random valid states, all clock models."""
    from mcmc_date_amd import synthetic as S

    topo = S.random_topology(129, seed=9)
    st = S.random_states(topo, 40, seed=9)
    rng = np.random.default_rng(9)
    birth, death, rvar = np.exp(0.3 * rng.standard_normal(40)), np.exp(0.3 * rng.standard_normal(40)), 0.2 + rng.random(40)
    cal = [M.Calibration("root", 0, 0.9, 0.025, 1.1, 0.025), M.Calibration("n", 5, 0.2, 0.025, None, 0.0)]
    con = [M.Constraint("k", 7, 3, 0.025)]
    for model in MODELS:
        pf = M.PriorFunction(1.0, model, cal, con, [], topo)
        lp = pf.logprior(M.StateBatch(st.heights, st.rates, st.time_height, st.rate_mean, birth, death, rvar))
        spec = O.PriorSpec(topo.parent, 1.0, model, [(0, 0.9, 0.025, 1.1, 0.025), (5, 0.2, 0.025, None, 0.0)], [(7, 3, 0.025)], [])
        ref = np.array([O.prior(spec, birth[b], death[b], st.time_height[b], st.heights[b], st.rate_mean[b], rvar[b], st.rates[b])[0]
                        for b in range(40)])
        assert close(lp, ref)
