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


def test_prior_gradient_large_tree_same_bits_whatever_the_batch(gpu):
    """The gradient kernel gives a chain 1 .. 4 waves (threads = nodes) depending on the batch; every per-node quantity is added
    up in the order of the one-wave form, so value and gradient of a chain are the same bits in a batch of 40 (four waves per
    chain), 700 (two) and 3000 (one), the value is the one of the value kernel, and central differences of the value agree."""
    from mcmc_date_amd import synthetic as S

    topo = S.random_topology(129, seed=9)
    B = 3000
    st = S.random_states(topo, B, seed=9)
    rng = np.random.default_rng(9)
    birth, death, rvar = np.exp(0.3 * rng.standard_normal(B)), np.exp(0.3 * rng.standard_normal(B)), 0.2 + rng.random(B)
    cal = [M.Calibration("root", 0, 0.9, 0.025, 1.1, 0.025), M.Calibration("n", 5, 0.2, 0.025, None, 0.0)]
    con = [M.Constraint("k", 7, 3, 0.025)]
    full = M.StateBatch(st.heights, st.rates, st.time_height, st.rate_mean, birth, death, rvar)
    for model in MODELS:
        pf = M.PriorFunction(1.0, model, cal, con, [], topo)
        lp, g = pf.grad(full)
        assert np.array_equal(lp, pf.logprior(full))
        for nb in (40, 700):
            lp_s, g_s = pf.grad(full.slice(0, nb))
            assert np.array_equal(lp_s, lp[:nb])
            for k in g_s:
                assert np.array_equal(np.asarray(g_s[k]), np.asarray(g[k])[:nb]), (model, nb, k)
        # one chain, a few coordinates, against differences of the value kernel
        b, eps = 3, 1e-6
        for v in (0, 1, 17, 128, 256):
            Hp, Hm = st.heights[b:b + 1].copy(), st.heights[b:b + 1].copy()
            Hp[0, v] += eps
            Hm[0, v] -= eps
            one = lambda Hx: pf.logprior(M.StateBatch(Hx, st.rates[b:b + 1], st.time_height[b:b + 1], st.rate_mean[b:b + 1], birth[b:b + 1],
                                                      death[b:b + 1], rvar[b:b + 1]))[0]
            fd = (one(Hp) - one(Hm)) / (2 * eps)
            gh = np.asarray(g["heights"])[b, v]
            assert abs(fd - gh) <= 1e-4 * max(1.0, abs(gh)), (model, v, fd, gh)


@pytest.mark.parametrize("name", ["12-leaves-variable-rate", "24-leaves-braces"])
@pytest.mark.parametrize("model", MODELS)
def test_prior_gradient_against_differences_of_the_oracle(gpu, golden, name, model):
    """Row f3, first part: d ln prior / d state from forward-mode duals on the device against central differences of
    the pinned value oracle (oracle/prior_oracle.c).  The prior is C^1 (soft bounds are piecewise quadratic), so
    fourth-order central differences with a relative step of 1e-4 are accurate to about 1e-7; tolerance 2e-5 * max(1, |g|)."""
    fx = golden[name]
    topo = M.Topology(fx["parent"])
    cal, con, br = tables(fx)
    ht = float(fx["prior_ht"])
    pf = M.PriorFunction(ht, model, cal, con, br, topo)
    spec = O.PriorSpec(fx["parent"], ht, model, [(c.node, c.lower, c.lower_p, c.upper, c.upper_p) for c in cal],
                       [(k.young, k.old, k.p) for k in con], [(b.nodes, b.sd) for b in br])
    s = batch(fx).slice(0, 6)
    # make sure some soft bounds are active: push the first calibrated node of chain 1 above its upper bound, and
    # spread the braced nodes of chain 2
    H = s.heights.copy()
    tHs = s.time_height.copy()
    if cal:
        c = cal[0]
        if c.upper is not None:
            tHs[1] = 1.3 * c.upper / H[1, c.node]
    s = M.StateBatch(H, s.rates, tHs, s.rate_mean, s.time_birth_rate, s.time_death_rate, s.rate_variance)
    lp, g = pf.grad(s)
    lp_ref = pf.logprior(s)
    assert np.array_equal(lp, lp_ref)                         # same value code
    n = topo.n_nodes

    def f(b, birth, death, tH, Hb, rMu, rVar, Rb):
        return O.prior(spec, birth, death, tH, Hb, rMu, rVar, Rb)[0]

    for b in range(6):
        base = [s.time_birth_rate[b], s.time_death_rate[b], s.time_height[b], s.heights[b].copy(), s.rate_mean[b], s.rate_variance[b], s.rates[b].copy()]
        assert np.isfinite(lp[b])
        near = abs(base[0] - base[1]) < 1e-6                   # the fixtures' chain 1 sits in the near-critical regime on purpose
        tol = 2e-4 if near else 2e-5                           # there: exact formulas at the regime's edge vs first-order value

        def at(i, v, x):
            a = list(base)
            if v is None:
                a[i] = x
            else:
                a[i] = base[i].copy()
                a[i][v] = x
            return f(b, *a)

        def stencil(i, v, x0):                                  # fourth-order central difference, step 1e-4 |x|
            h = 1e-4 * abs(x0)
            return (-at(i, v, x0 + 2 * h) + 8 * at(i, v, x0 + h) - 8 * at(i, v, x0 - h) + at(i, v, x0 - 2 * h)) / (12 * h)

        def fd_scalar(i):
            return stencil(i, None, base[i])

        for i, key in ((0, "time_birth_rate"), (1, "time_death_rate"), (2, "time_height"), (4, "rate_mean"), (5, "rate_variance")):
            fd = fd_scalar(i)
            if near and i in (0, 1):
                continue                                       # the value has a kink of order 1e-6 across the regime's edge
            assert abs(g[key][b] - fd) <= tol * max(1.0, abs(fd)), (key, b, g[key][b], fd)
        assert g["rate_mean"][b] == -ht and g["rates"][b, 0] == 0.0
        for v in range(1, n):
            for i, key in ((3, "heights"), (6, "rates")):
                x0 = base[i][v]
                if x0 == 0.0:
                    continue                                   # leaves: heights are not free parameters (mask)
                fd = stencil(i, v, x0)
                assert abs(g[key][b, v] - fd) <= tol * max(1.0, abs(fd)), (key, b, v, g[key][b, v], fd)
    # outside the support the gradient is NaN; in the near-critical regime it is finite and continuous across the edge
    bad = batch(fx).slice(0, 3)
    R = bad.rates.copy(); R[0, 3] = -1.0
    birth = bad.time_birth_rate.copy(); death = bad.time_death_rate.copy()
    birth[1] = death[1] + 1e-9
    birth[2] = death[2] = 1.0
    lp2, g2 = pf.grad(M.StateBatch(bad.heights, R, bad.time_height, bad.rate_mean, birth, death, bad.rate_variance))
    assert lp2[0] == -np.inf and np.all(np.isnan(g2["heights"][0])) and np.isnan(g2["rate_variance"][0])
    assert np.isfinite(lp2[1]) and np.all(np.isfinite(g2["heights"][1])) and np.isfinite(g2["time_birth_rate"][1])
    birth[1] = death[1] + 2e-6                                 # just outside the regime: the exact formulas
    _, g3 = pf.grad(M.StateBatch(bad.heights, R, bad.time_height, bad.rate_mean, birth, death, bad.rate_variance))
    assert np.allclose(g2["heights"][1], g3["heights"][1], rtol=1e-4, atol=1e-6)
    assert np.isfinite(lp2[2]) and np.all(np.isfinite(g2["heights"][2])) and np.isfinite(g2["time_birth_rate"][2])


def test_hamiltonian_target_gradient_in_position_layout(gpu, golden):
    """ln [prior x likelihood x jacobianRootBranch] (htargetWith, app/Hamiltonian.hs:72-92) and its gradient in the
    masked, reversed position vector of toVector (:49-53): directional derivative against differences of the oracles."""
    fx = golden["24-leaves-braces"]
    topo = M.Topology(fx["parent"])
    cal, con, br = tables(fx)
    ht = float(fx["prior_ht"])
    pf = M.PriorFunction(ht, "UncorrelatedGamma", cal, con, br, topo)
    lik = M.MvnLikelihood(M.Full(fx["mu"], fx["sigma_inv"], float(fx["logdet"]))).bind_tree(topo)
    spec = O.PriorSpec(fx["parent"], ht, "UncorrelatedGamma", [(c.node, c.lower, c.lower_p, c.upper, c.upper_p) for c in cal],
                       [(k.young, k.old, k.p) for k in con], [(b.nodes, b.sd) for b in br])
    s = batch(fx).slice(2, 6)
    mask = M.get_mask(True, topo)
    val, grad = M.target_grad(mask, lik, pf, s)
    assert grad.shape == (4, 73)                                  # 73 positions for the 24-leaf dataset

    def target(x: M.State) -> float:
        lp = O.prior(spec, x.time_birth_rate, x.time_death_rate, x.time_height, x.time_tree, x.rate_mean, x.rate_variance, x.rate_tree)[0]
        ll, lj = O.tree_loglik_full_batch(fx["parent"], x.time_tree[None], x.rate_tree[None], np.array([x.time_height]), np.array([x.rate_mean]),
                                          fx["mu"], fx["sigma_inv"], float(fx["logdet"]))
        return lp + ll[0] + lj[0]

    rng = np.random.default_rng(0)
    for b in range(4):
        x = M.State(s.time_birth_rate[b], s.time_death_rate[b], s.time_height[b], s.heights[b], s.rate_mean[b], s.rate_variance[b], s.rates[b])
        assert abs(val[b] - target(x)) <= 1e-9 * max(1.0, abs(val[b]))
        q = M.to_vector(mask, x)
        for _ in range(3):
            d = rng.normal(size=q.shape) * np.abs(q)
            h = 1e-5
            f = lambda a: target(M.from_vector_with(mask, x, q + a * h * d))
            fd = (-f(2) + 8 * f(1) - 8 * f(-1) + f(-2)) / (12 * h)
            an = float(grad[b] @ d)
            assert abs(an - fd) <= 1e-5 * max(1.0, abs(fd)), (b, an, fd)


def test_device_leapfrog(gpu, golden):
    """Row f3, second part: the leapfrog integrator on the device.  Position, value and gradient agree with
    hamiltonian.target_grad; the map is reversible (forward, flip the momenta, forward again returns to the start); the
    energy error falls fourfold when the step is halved (second order); direction -1 equals flipped momenta."""
    fx = golden["24-leaves-braces"]
    topo = M.Topology(fx["parent"])
    cal, con, br = tables(fx)
    pf = M.PriorFunction(float(fx["prior_ht"]), "UncorrelatedGamma", cal, con, br, topo)
    lik = M.MvnLikelihood(M.Full(fx["mu"], fx["sigma_inv"], float(fx["logdet"]))).bind_tree(topo)
    B = 8
    # typical posterior states: a short Metropolis-Hastings run from the initial state
    ps, _ = M.proposals(topo, br, calibrations_available=True)
    smp = M.Sampler(lik, pf, ps, B, seed=2)
    x0 = M.init_with(topo, fx["mean_lengths"])
    x0.time_height = float(fx["prior_ht"])
    smp.set_initial_state(x0)
    for period in (50, 50, 100, 100):
        smp.run(period)
        smp.autotune()
    s = smp.state()
    lf = M.Leapfrog(lik, pf, True, B)
    assert lf.dim == 73
    lf.set_state(s)
    q0, v0, g0 = lf.position()
    mask = M.get_mask(True, topo)
    val, grad = M.target_grad(mask, lik, pf, s)
    assert np.allclose(v0, val, rtol=1e-13) and np.allclose(g0, grad, rtol=1e-11, atol=1e-9)
    for b in range(B):
        x = M.State(s.time_birth_rate[b], s.time_death_rate[b], s.time_height[b], s.heights[b], s.rate_mean[b], s.rate_variance[b], s.rates[b])
        assert np.array_equal(q0[b], M.to_vector(mask, x))
    rng = np.random.default_rng(3)
    inv_mass = np.full(lf.dim, 1.0)
    scale = 1.0 / np.maximum(1.0, np.abs(g0).max(axis=1))            # keep the trajectories inside the support
    p0 = rng.normal(size=(B, lf.dim))
    eps = 2e-2 * scale
    # reversibility
    p1 = lf.leapfrog(p0, eps, inv_mass, 12)
    q1, v1, _ = lf.position()
    assert np.all(np.isfinite(v1)) and np.all(np.abs(q1 - q0).max(axis=1) > 1e-6)
    p2 = lf.leapfrog(-p1, eps, inv_mass, 12)
    q2, v2, _ = lf.position()
    assert np.allclose(q2, q0, rtol=1e-10, atol=1e-12) and np.allclose(-p2, p0, rtol=1e-8, atol=1e-10) and np.allclose(v2, v0, rtol=1e-11)
    # direction -1 from the start equals integrating with flipped momenta
    lf.set_state(s)
    pb = lf.leapfrog(p0, eps, inv_mass, 5, direction=-np.ones(B))
    qb, _, _ = lf.position()
    lf.set_state(s)
    pf2 = lf.leapfrog(-p0, eps, inv_mass, 5)
    qf, _, _ = lf.position()
    assert np.allclose(qb, qf, rtol=1e-12, atol=0) and np.allclose(pb, -pf2, rtol=1e-12, atol=1e-14)
    # second order: energy error over a fixed time T = n eps
    def energy_error(n):
        lf.set_state(s)
        pn = lf.leapfrog(p0, eps * (16.0 / n), inv_mass, n)
        _, vn, _ = lf.position()
        return np.abs((-vn + 0.5 * np.sum(pn * pn, axis=1)) - (-v0 + 0.5 * np.sum(p0 * p0, axis=1)))
    e16, e32 = energy_error(16), energy_error(32)
    ratio = e16 / e32
    assert np.all(e16 < 0.5) and np.median(ratio) > 3.0 and np.median(ratio) < 5.0, (e16, e32)


def test_hmc_chains_agree_with_metropolis_hastings_chains(gpu, golden):
    """End to end: fixed-length HMC transitions on the device leapfrog sample ln [prior x likelihood x
    jacobianRootBranch]; Metropolis-Hastings chains in which EVERY proposal carries the root-branch lift have the same
    target.  Node-age means of the two samplers on tests/12-leaves-variable-rate agree within 2 %."""
    import dataclasses

    from mcmc_date_amd import monitor as MO

    fx = golden["12-leaves-variable-rate"]
    topo = M.Topology(fx["parent"])
    cal, con, br = tables(fx)
    ht = float(fx["prior_ht"])
    pf = M.PriorFunction(ht, "UncorrelatedGamma", cal, con, br, topo)
    lik = M.MvnLikelihood(M.Full(fx["mu"], fx["sigma_inv"], float(fx["logdet"]))).bind_tree(topo)
    B = 64
    ps, _ = M.proposals(topo, br, calibrations_available=True, exact_jacobians=True)
    ps = [dataclasses.replace(p, jac_root=True) for p in ps]
    smp = M.Sampler(lik, pf, ps, B, seed=77)
    x0 = M.init_with(topo, fx["mean_lengths"])
    x0.time_height = ht
    smp.set_initial_state(x0)
    smp.burn_in(fast=[10, 10, 20, 40, 80], slow=[100, 200, 300, 400])
    tr = MO.collect(smp, 3000, period=50, accumulate=True)
    ages_mh = smp.node_age_summary()[0]
    mask = M.get_mask(True, topo)
    qs = np.array([M.to_vector(mask, M.State(tr.time_birth_rate[k, b], tr.time_death_rate[k, b], tr.time_height[k, b], tr.heights[k, b],
                                             tr.rate_mean[k, b], tr.rate_variance[k, b], tr.rates[k, b]))
                   for k in range(tr.heights.shape[0]) for b in range(0, B, 4)])
    inv_mass = qs.var(axis=0)                                   # masses = inverse posterior variances
    lf = M.Leapfrog(lik, pf, True, B)
    lf.set_state(smp.state())
    rng = np.random.default_rng(5)
    acc, ages, n_tr = [], np.zeros(topo.n_nodes), 1500
    for it in range(n_tr):
        a = M.hmc_transition(lf, rng, 0.04, inv_mass, 12)
        acc.append(a.mean())
        s = lf.state()
        ages += (s.time_height[:, None] * s.heights).mean(axis=0)
    ages_hmc = ages / n_tr
    inner = ~topo.leaves
    rel = np.abs(ages_hmc[inner] - ages_mh[inner]) / ages_mh[inner]
    assert 0.5 < np.mean(acc) <= 1.0, np.mean(acc)
    assert rel.max() <= 0.02, (rel, np.mean(acc))


def test_nuts_chains_agree_with_metropolis_hastings_chains(gpu, golden):
    """NUTS (Hoffman & Gelman 2014, Algorithm 3) with dual-averaging step sizes (Algorithm 6) on the device leapfrog
    against Metropolis-Hastings chains with the same target (every proposal lifted with jacobianRootBranch): node-age
    means on tests/12-leaves-variable-rate within 3 %; the adapted step size gives the target acceptance statistic."""
    import dataclasses

    from mcmc_date_amd import monitor as MO

    fx = golden["12-leaves-variable-rate"]
    topo = M.Topology(fx["parent"])
    cal, con, br = tables(fx)
    ht = float(fx["prior_ht"])
    pf = M.PriorFunction(ht, "UncorrelatedGamma", cal, con, br, topo)
    lik = M.MvnLikelihood(M.Full(fx["mu"], fx["sigma_inv"], float(fx["logdet"]))).bind_tree(topo)
    B = 32
    ps, _ = M.proposals(topo, br, calibrations_available=True, exact_jacobians=True)
    ps = [dataclasses.replace(p, jac_root=True) for p in ps]
    smp = M.Sampler(lik, pf, ps, B, seed=78)
    x0 = M.init_with(topo, fx["mean_lengths"])
    x0.time_height = ht
    smp.set_initial_state(x0)
    smp.burn_in(fast=[10, 10, 20, 40, 80], slow=[100, 200, 300, 400])
    tr = MO.collect(smp, 4000, period=50, accumulate=True)
    ages_mh = smp.node_age_summary()[0]
    mask = M.get_mask(True, topo)
    qs = np.array([M.to_vector(mask, M.State(tr.time_birth_rate[k, b], tr.time_death_rate[k, b], tr.time_height[k, b], tr.heights[k, b],
                                             tr.rate_mean[k, b], tr.rate_variance[k, b], tr.rates[k, b]))
                   for k in range(tr.heights.shape[0]) for b in range(0, B, 2)])
    inv_mass = qs.var(axis=0)
    lf = M.Leapfrog(lik, pf, True, B)
    lf.set_state(smp.state())
    rng = np.random.default_rng(6)
    da = M.DualAveraging(np.full(B, 0.03), delta=0.65)
    eps = np.full(B, 0.03)
    for _ in range(120):                                       # warm-up: step sizes only (masses come from the MH sample)
        alpha, depth = M.nuts_transition(lf, rng, eps, inv_mass, max_depth=6)
        eps = da.update(alpha)
    eps = da.final()
    assert np.all((eps > 0.005) & (eps < 0.5)), eps
    ages, alphas, depths, n_tr = np.zeros(topo.n_nodes), [], [], 350
    for _ in range(n_tr):
        alpha, depth = M.nuts_transition(lf, rng, eps, inv_mass, max_depth=6)
        alphas.append(alpha.mean())
        depths.append(depth.mean())
        s = lf.state()
        ages += (s.time_height[:, None] * s.heights).mean(axis=0)
    ages_nuts = ages / n_tr
    inner = ~topo.leaves
    rel = np.abs(ages_nuts[inner] - ages_mh[inner]) / ages_mh[inner]
    assert 0.45 < np.mean(alphas) < 0.9 and 1.5 < np.mean(depths) <= 6.0, (np.mean(alphas), np.mean(depths))
    assert rel.max() <= 0.03, (rel, np.mean(alphas), np.mean(depths))   # Monte Carlo error of 32 chains x 350 transitions: ~1 %
    # warm-up of step sizes AND diagonal masses without outside knowledge, from states the Metropolis-Hastings cycle has
    # brought to the posterior (the reference runs NUTS as one proposal of that cycle): the masses it finds are the
    # posterior variances of the Metropolis-Hastings sample, typically within a factor 1.6
    lf2 = M.Leapfrog(lik, pf, True, B)
    lf2.set_state(smp.state())
    eps_w, inv_mass_w = M.nuts_warmup(lf2, np.random.default_rng(8), n_windows=4, window=50)
    ratio = inv_mass_w / inv_mass
    assert np.median(np.abs(np.log(ratio))) < np.log(1.6) and np.mean(np.abs(np.log(ratio)) < np.log(3.0)) > 0.9, ratio
    alpha, _ = M.nuts_transition(lf2, np.random.default_rng(9), eps_w, inv_mass_w, max_depth=6)
    assert 0.3 < alpha.mean() <= 1.0
    # the reference's --hamiltonian mode: the Metropolis-Hastings cycle plus one NUTS proposal per iteration
    a_mean, ages_mix = M.run_cycle_with_nuts(smp, lf2, np.random.default_rng(10), 150, eps_w, inv_mass_w, accumulate=True)
    rel_mix = np.abs(ages_mix[inner] - ages_mh[inner]) / ages_mh[inner]
    assert 0.3 < a_mean <= 1.0 and rel_mix.max() <= 0.04, (a_mean, rel_mix)
    # leapfrog throughput of the device integrator (reported in DESIGN.md)
    import time
    lf2.leapfrog(np.zeros((B, lf2.dim)), 1e-3, inv_mass_w, 10)
    t0 = time.perf_counter()
    lf2.leapfrog(np.zeros((B, lf2.dim)), 1e-3, inv_mass_w, 200)
    print("leapfrog: %.1f us per step for %d chains" % (1e6 * (time.perf_counter() - t0) / 200, B))


def test_hmc_handle_errors(gpu, golden):
    """Structural faults of the leapfrog entry points come back as error codes with messages (never a crash)."""
    import ctypes as C

    fx = golden["06-leaves-constant-rate"]
    topo = M.Topology(fx["parent"])
    pf = M.PriorFunction(float(fx["prior_ht"]), "UncorrelatedGamma", [], [], [], topo)
    lik = M.MvnLikelihood(M.Full(fx["mu"], fx["sigma_inv"], float(fx["logdet"]))).bind_tree(topo)
    lf = M.Leapfrog(lik, pf, False, 3)
    assert lf.dim == 2 + (topo.n_nodes - int(topo.leaves.sum()) - 1) + 2 + (topo.n_nodes - 1)      # no tH without calibrations
    with pytest.raises(M.McdError, match="set_state"):
        lf.position()
    with pytest.raises(M.McdError, match="set_state"):
        lf.leapfrog(np.zeros((3, lf.dim)), 0.01, 1.0, 1)
    x0 = M.init_with(topo, fx["mean_lengths"])
    lf.set_state(M.StateBatch.from_states([x0] * 3))
    q, v, g = lf.position()
    assert np.all(np.isfinite(v)) and np.all(np.isfinite(g))            # birth = death = 1: the near-critical start has a gradient
    with pytest.raises(M.McdError, match="positive"):
        lf.leapfrog(np.zeros((3, lf.dim)), 0.01, np.zeros(lf.dim), 1)
    with pytest.raises(ValueError):
        lf.leapfrog(np.zeros((2, lf.dim)), 0.01, 1.0, 1)
    # a step that leaves the support: NaN value, the caller rejects
    p = np.zeros((3, lf.dim))
    p[:, -1] = -1e6                                                       # drives the birth rate far below zero
    lf.leapfrog(p, 1e-3, 1.0, 1)
    _, v2, _ = lf.position()
    assert not np.any(np.isfinite(v2))
    other = M.Topology(np.array([-1, 0, 1, 1, 0], np.int32))
    pf_other = M.PriorFunction(1.0, "UncorrelatedGamma", [], [], [], other)
    with pytest.raises(M.McdError):
        M.Leapfrog(lik, pf_other, False, 3)
