"""GPU parity of the sparse form (mcd_sparse_*, csrc/k_sparse.hip: the precision matrix in CSR on the device, no densification)
against the oracle's restatement of logDensitySparseMultivariateNormal (app/Probability.hs:178-184; oracle.logpdf_sparse) and
against scipy.sparse: the reference's own sparse operands (the graphical-lasso estimate of the 7-taxon mtCDNApri analysis,
prepare's SparseS record), a synthetic 1007-taxon problem (N = 2011: the size of the reference's large example,
tutorial/main/tutorial.org:487-496) and every tile geometry up to N = 8192.  Tolerance: 1e-12 relative on the quadratic form
(fixed summation order on both sides, different orders: rounding only)."""
import numpy as np
import pytest
import scipy.sparse as sps

import mcmc_date_amd as M
import oracle as O

pytestmark = pytest.mark.gpu


def banded_random_precision(n, seed, band=3, extra=4):
    from mcmc_date_amd import synthetic as S

    return S.banded_precision(n, seed, band, extra)


def test_reference_sparse_operands_mtcdnapri(gpu):
    """`prepare ... "SparseMultivariateNormal 0.1"` on the reference's mtCDNApri inputs (N = 11, 87 of 121 entries): raw vectors
    against the oracle's sparse restatement and against the densified handle, states against the oracle's tree likelihood."""
    import mtcdnapri as A

    an = A.analysis("SparseMultivariateNormal 0.1")
    lhd = an.prep.lhd
    sp = M.SparseLikelihood(lhd)
    assert sp.n == 11 and sp.nnz == len(lhd.sigma_inv_assoc) < 121
    ii = np.array([ij[0] for ij, _ in lhd.sigma_inv_assoc]); jj = np.array([ij[1] for ij, _ in lhd.sigma_inv_assoc])
    vv = np.array([v for _, v in lhd.sigma_inv_assoc])
    rng = np.random.default_rng(1)
    X = np.asarray(lhd.mu) * np.exp(0.3 * rng.standard_normal((40, 11)))
    ll = sp.logpdf(X)
    ref = np.array([O.logpdf_sparse(lhd.mu, ii, jj, vv, lhd.logdet_sigma, x) for x in X])
    assert np.max(np.abs(ll - ref) / np.maximum(1.0, np.abs(ref))) <= 1e-12
    dense = M.MvnLikelihood(lhd)
    assert np.max(np.abs(ll - dense.logpdf(X)) / np.maximum(1.0, np.abs(ref))) <= 1e-10
    ll_g, G = sp.grad(X)
    assert np.allclose(ll_g, ll, rtol=1e-13, atol=0)        # (the gradient call takes the row form, the value call the one-launch form: rounding)
    assert np.allclose(G, -(an.sigma_inv @ (X - lhd.mu).T).T, rtol=1e-12, atol=1e-12 * np.abs(G).max())
    # states
    topo = an.topo
    st = M.StateBatch.from_states([M.init_with(topo, an.prep.mean_lengths)] * 5)
    st.time_height = np.array([15.0, 17.0, 19.0, 21.0, 30.0])
    st.rate_mean = np.full(5, 0.004)
    st.rates = st.rates * np.exp(0.2 * rng.standard_normal(st.rates.shape))
    llt, lj = sp.bind_tree(topo).loglik(st)
    reft, refj = O.tree_loglik_full_batch(topo.parent, st.heights, st.rates, st.time_height, st.rate_mean, an.mu, an.sigma_inv, an.logdet)
    assert np.max(np.abs(llt - reft) / np.maximum(1.0, np.abs(reft))) <= 1e-11 and np.allclose(lj, refj, rtol=1e-13, atol=0)


def test_thousand_taxa(gpu):
    """N = 2011 (1007 leaves): 8 chains per tile, 37 chains (a ragged last tile), device-resident; ll against the oracle's sparse
    restatement, the gradient against scipy.sparse, states against distances from the oracle."""
    import torch

    from mcmc_date_amd import synthetic as S

    topo = S.random_topology(1007, seed=5)
    n = topo.n_nodes - 2
    assert n == 2011
    P, assoc = banded_random_precision(n, seed=5)
    rng = np.random.default_rng(6)
    mu = rng.uniform(0.01, 0.2, n)
    logdet = -float(np.linalg.slogdet(P.toarray())[1])       # log det Sigma = -log det P
    sp = M.SparseLikelihood(M.Sparse(mu, assoc, logdet))
    assert sp.nnz == P.nnz
    B = 37
    X = mu * np.exp(0.2 * rng.standard_normal((B, n)))
    Xd = torch.as_tensor(X, device=gpu)
    ll = sp.logpdf(Xd).cpu().numpy()
    dx = X - mu
    q = np.einsum("bi,bi->b", dx, (P @ dx.T).T)
    ref = -n * 0.9189385332046727 - 0.5 * (logdet + q)
    assert np.max(np.abs(ll - ref) / np.abs(ref)) <= 1e-12
    coo = P.tocoo()
    for b in (0, 17, 36):
        assert abs(ll[b] - O.logpdf_sparse(mu, coo.row, coo.col, coo.data, logdet, X[b])) <= 1e-12 * abs(ref[b])
    ll2, G = sp.grad(Xd)
    assert np.allclose(ll2.cpu().numpy(), ll, rtol=1e-13, atol=0)     # (row form beside the one-launch form: another order of summation)
    Gref = -(P @ dx.T).T
    assert np.max(np.abs(G.cpu().numpy() - Gref)) <= 1e-12 * np.abs(Gref).max()
    assert np.array_equal(sp.logpdf(X), ll)                  # host pointers: the same bits
    # tree states
    st = S.random_states(topo, B, seed=7)
    tl = sp.bind_tree(topo)
    llt, lj = tl.loglik(st.to(gpu))
    D = np.array([O.distances(topo.parent, st.heights[b], st.rates[b], st.time_height[b], st.rate_mean[b]) for b in range(B)])
    dd = D - mu
    reft = -n * 0.9189385332046727 - 0.5 * (logdet + np.einsum("bi,bi->b", dd, (P @ dd.T).T))
    assert np.max(np.abs(llt.cpu().numpy() - reft) / np.abs(reft)) <= 1e-11
    assert np.allclose(lj.cpu().numpy(), np.log(1.0 / D[:, 0]), rtol=1e-13, atol=0)
    # NaN in one chain stays in that chain
    Xn = X.copy()
    Xn[3, 100] = np.nan
    lln = sp.logpdf(Xn)
    assert np.isnan(lln[3]) and np.array_equal(np.delete(lln, 3), np.delete(ll, 3))


@pytest.mark.parametrize("n,chains_per_tile", [(64, 16), (1100, 16), (1500, 8), (3000, 4), (5000, 2), (8192, 1)])
def test_every_tile_geometry(gpu, n, chains_per_tile):
    P, assoc = banded_random_precision(n, seed=n, band=2, extra=2)
    rng = np.random.default_rng(n + 1)
    mu = rng.uniform(0.01, 0.2, n)
    sp = M.SparseLikelihood(M.Sparse(mu, assoc, 12.5))
    B = 2 * chains_per_tile + 1
    X = mu + 0.01 * rng.standard_normal((B, n))
    dx = X - mu
    ref = -n * 0.9189385332046727 - 0.5 * (12.5 + np.einsum("bi,bi->b", dx, (P @ dx.T).T))
    ll, G = sp.grad(X)
    assert np.max(np.abs(ll - ref) / np.abs(ref)) <= 1e-12
    Gref = -(P @ dx.T).T
    assert np.max(np.abs(G - Gref)) <= 1e-12 * np.abs(Gref).max()


def test_structural_faults_and_duplicates(gpu):
    mu = np.array([0.1, 0.2, 0.3])
    # entries of one position add up (an association list may repeat a position); an empty row is fine
    sp = M.SparseLikelihood(M.Sparse(mu, [((0, 0), 1.0), ((0, 0), 1.5), ((1, 1), 2.0), ((0, 1), 0.5), ((1, 0), 0.5)], 0.7))
    assert sp.nnz == 4
    x = np.array([[0.3, 0.1, 5.0]])
    dx = x[0] - mu
    Pd = np.array([[2.5, 0.5, 0.0], [0.5, 2.0, 0.0], [0.0, 0.0, 0.0]])
    assert abs(sp.logpdf(x)[0] - (-3 * 0.9189385332046727 - 0.5 * (0.7 + dx @ Pd @ dx))) <= 1e-14
    with pytest.raises(M.McdError):
        M.SparseLikelihood(M.Sparse(mu, [((0, 3), 1.0)], 0.0))
    with pytest.raises(M.McdError):
        M.SparseLikelihood(M.Sparse(mu, [((0, 0), np.nan)], 0.0))
    with pytest.raises(M.McdError):
        M.SparseLikelihood(M.Sparse(np.zeros(M.likelihood.MAX_SPARSE_DIM + 1), [((0, 0), 1.0)], 0.0))
    with pytest.raises(M.RootNotBifurcating):
        sp1 = M.SparseLikelihood(M.Sparse(np.array([0.1, 0.2]), [((0, 0), 1.0), ((1, 1), 1.0)], 0.0))
        sp1.bind_tree(M.Topology(np.array([-1, 0, 0, 0], dtype=np.int32)))
    with pytest.raises(M.McdError):
        sp.bind_tree(M.Topology(np.array([-1, 0, 1, 1, 0, 4, 4], dtype=np.int32)))     # 7 nodes: dimension 5, not 3


def _sparse_problem(n_leaves, B):
    from mcmc_date_amd import synthetic as S

    topo = S.random_topology(n_leaves, seed=9)
    n = topo.n_nodes - 2
    P, assoc = banded_random_precision(n, seed=n)
    s0 = S.random_states(topo, B, seed=11)
    s0.time_birth_rate = np.full(B, 1.0); s0.time_death_rate = np.full(B, 0.8); s0.rate_variance = np.full(B, 0.3)
    # the mean vector near the states' distances, so that ln likelihoods are of the size a sampler sees
    D = np.array([O.distances(topo.parent, s0.heights[b], s0.rates[b], s0.time_height[b], s0.rate_mean[b]) for b in range(B)])
    mu = D.mean(axis=0)
    Pd = P.toarray()
    logdet = -float(np.linalg.slogdet(Pd)[1])
    return topo, P, Pd, assoc, mu, logdet, s0


@pytest.mark.parametrize("n_leaves,B,n_steps", [(7, 8, 700), (40, 16, 600), (200, 16, 200), (513, 64, 300), (1007, 8, 60), (1007, 512, 600)])
def test_metropolis_hastings_over_a_sparse_likelihood(gpu, n_leaves, B, n_steps):
    """The lock-step driver over a likelihood whose precision matrix stays sparse on the device (mcd_mh_create_sparse): the
    reference's production configuration -- `mhg` with likelihoodFunction (Sparse ...), app/Main.hs:474, 257-277; app/Probability.hs:
    178-184 -- from a 13-node tree (every proposal inside a segment: all distances fit the list) over 399 and 1025 nodes (two chains per
    workgroup) to the size of its 1007-taxon example (2013 nodes, N = 2011, one chain per workgroup; 512 chains x 600 lock steps, two
    recomputations of q).  Segments (k_mh_segment_sparse.hip): the quadratic form updated through the rows of the moved distances.
    Step by step against the CPU twin, which evaluates the same precision matrix densely at every step: identical accept / reject
    decisions, ln acceptance ratios within the twin's tolerance, final states within 1e-9."""
    topo, P, Pd, assoc, mu, logdet, s0 = _sparse_problem(n_leaves, B)
    sp = M.SparseLikelihood(M.Sparse(mu, assoc, logdet))
    tl = sp.bind_tree(topo)
    pf = M.PriorFunction(1.0, "UncorrelatedGamma", [], [], [], topo)
    ps, _ = M.proposals(topo, [], calibrations_available=True)
    smp = M.Sampler(tl, pf, ps, B, seed=13)
    smp.set_state(s0)
    spec = O.PriorSpec(topo.parent, 1.0, "UncorrelatedGamma", [], [], [])
    twin = O.MhChains(O.MhModel(topo.parent, mu, Pd, logdet, spec, M.table_arrays(ps)), s0.time_birth_rate, s0.time_death_rate,
                      s0.time_height, s0.heights, s0.rate_mean, s0.rate_variance, s0.rates, seed=13)
    post0 = smp.posterior()
    assert np.allclose(post0[:, 1], twin_ll(topo, s0, mu, P, logdet), rtol=1e-11)
    cyc = M.cycle_schedule(ps, 1, np.random.default_rng(0))
    sched = np.tile(cyc, (1, n_steps // cyc.shape[1] + 1))[:, :n_steps]
    tol = 1e-8 + 1e-12 * np.abs(post0[:, :2]).max()
    ta, tk = smp.run_schedule(sched, trace=True)
    assert "segments over a sparse precision matrix" in smp.last_path()
    ra, rk = twin.run(sched, trace=True)
    fin = np.isfinite(ra)
    assert np.array_equal(np.isfinite(ta), fin) and np.all(np.abs(ta[fin] - ra[fin]) <= tol + 1e-10 * np.abs(ra[fin]))
    assert np.array_equal(tk, rk) and 0.02 < tk.mean() < 0.98
    s = smp.state()
    for a, b in ((s.time_height, twin.tH), (s.heights, twin.H), (s.rate_mean, twin.rMu), (s.rates, twin.R)):
        assert np.allclose(a, b, rtol=1e-9, atol=0)
    assert np.allclose(smp.posterior(), twin.post, rtol=1e-11, atol=tol)


@pytest.mark.parametrize("n_leaves,B,n_steps", [(7, 5, 600), (200, 33, 700), (1007, 6, 300)])
def test_sparse_segments_against_a_full_product_at_every_step(gpu, n_leaves, B, n_steps, knobs):
    """The same chains with the segments (incremental quadratic form, q recomputed every 256 steps) and with round 3's structure -- the
    step kernel and a full product at every step (knob MCD_MH_SEGMENTS = 0): identical decisions, states, ln priors and ln Jacobians;
    ln likelihoods to rounding.  An odd batch (a workgroup with one chain missing); runs continued by the other structure."""
    topo, P, Pd, assoc, mu, logdet, s0 = _sparse_problem(n_leaves, B)
    tl = M.SparseLikelihood(M.Sparse(mu, assoc, logdet)).bind_tree(topo)
    pf = M.PriorFunction(1.0, "UncorrelatedGamma", [], [], [], topo)
    ps, _ = M.proposals(topo, [], calibrations_available=True)
    cyc = M.cycle_schedule(ps, 1, np.random.default_rng(1))
    sched = np.tile(cyc, (1, n_steps // cyc.shape[1] + 1))[:, :n_steps]
    out = {}
    for seg in ("1", "0"):
        knobs.setenv("MCD_MH_SEGMENTS", seg)
        smp = M.Sampler(tl, pf, ps, B, seed=21)
        smp.set_state(s0)
        ta, tk = smp.run_schedule(sched, trace=True)
        assert ("segments over a sparse" in smp.last_path()) == (seg == "1") and ("sparse product" in smp.last_path()) == (seg == "0")
        knobs.setenv("MCD_MH_SEGMENTS", "0" if seg == "1" else "1")      # ... and continued by the other structure
        tb, tkb = smp.run_schedule(sched[:, :97], trace=True)
        out[seg] = (ta, tk, tb, tkb, smp.state(), smp.posterior())
    a, b = out["1"], out["0"]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
    for x, y in ((a[0], b[0]), (a[2], b[2])):
        fin = np.isfinite(y)
        assert np.array_equal(np.isfinite(x), fin) and np.all(np.abs(x[fin] - y[fin]) <= 1e-9 + 1e-11 * np.abs(y[fin]))
    for f in ("time_birth_rate", "time_death_rate", "time_height", "heights", "rate_mean", "rate_variance", "rates"):
        assert np.array_equal(getattr(a[4], f), getattr(b[4], f)), f
    assert np.array_equal(a[5][:, 0], b[5][:, 0]) and np.array_equal(a[5][:, 2], b[5][:, 2])
    assert np.allclose(a[5][:, 1], b[5][:, 1], rtol=1e-11, atol=1e-9)


def test_one_launch_form_against_the_row_form(gpu, knobs):
    """The one-launch quadratic form (k_sparse_quad: a workgroup per one or two chains, the flat entry stream; upper triangle for a
    symmetric matrix) against the three-launch row form (lanes = chains) and scipy.sparse: symmetric and non-symmetric matrices, batches
    that take one and two chains per workgroup, tree states; the gradient of a non-symmetric matrix is -1/2 (P + P^T) dx."""
    import torch

    from mcmc_date_amd import synthetic as S

    rng = np.random.default_rng(3)
    for n, B, symmetric in ((11, 5, True), (255, 700, True), (2011, 513, True), (300, 64, False), (8001, 3, True)):
        P, assoc = banded_random_precision(n, seed=n)
        if not symmetric:
            assoc = [((i, j), v * (1.3 if i < j else 1.0)) for (i, j), v in assoc]
            P = sps.coo_matrix(([v for _, v in assoc], ([ij[0] for ij, _ in assoc], [ij[1] for ij, _ in assoc])), shape=(n, n)).tocsr()
        mu = rng.uniform(0.01, 0.2, n)
        sp = M.SparseLikelihood(M.Sparse(mu, assoc, 0.3))
        X = mu + 0.01 * rng.standard_normal((B, n))
        dx = X - mu
        ref = -n * 0.9189385332046727 - 0.5 * (0.3 + np.einsum("bi,bi->b", dx, (P @ dx.T).T))
        got = {}
        for form in ("1", "0"):
            knobs.setenv("MCD_SPARSE_QUAD", form)
            got[form] = sp.logpdf(X)
            assert np.max(np.abs(got[form] - ref) / np.abs(ref)) <= 1e-12, (n, B, form)
            Xd = torch.as_tensor(X, device="cuda")
            assert np.array_equal(sp.logpdf(Xd).cpu().numpy(), got[form])      # device-resident = host-pointer, bit for bit
        knobs.delenv("MCD_SPARSE_QUAD")
        ll_g, G = sp.grad(X)
        Ps = 0.5 * (P + P.T)
        assert np.allclose(G, -(Ps @ dx.T).T, rtol=1e-12, atol=1e-13 * np.abs(G).max())
        assert np.max(np.abs(ll_g - ref) / np.abs(ref)) <= 1e-12
    # tree states through the one-launch form
    topo = S.random_topology(129, seed=4)
    n = topo.n_nodes - 2
    P, assoc = banded_random_precision(n, seed=n)
    mu = rng.uniform(0.01, 0.2, n)
    st = S.random_states(topo, 70, seed=5)
    tl = M.SparseLikelihood(M.Sparse(mu, assoc, 0.0)).bind_tree(topo)
    res = {}
    for form in ("1", "0"):
        knobs.setenv("MCD_SPARSE_QUAD", form)
        res[form] = tl.loglik(st)
    assert np.allclose(res["1"][0], res["0"][0], rtol=1e-12) and np.array_equal(res["1"][1], res["0"][1])
    reft, refj = O.tree_loglik_full_batch(topo.parent, st.heights, st.rates, st.time_height, st.rate_mean, mu, P.toarray(), 0.0)
    assert np.max(np.abs(res["1"][0] - reft) / np.maximum(1.0, np.abs(reft))) <= 1e-11 and np.allclose(res["1"][1], refj, rtol=1e-13, atol=0)


def twin_ll(topo, st, mu, P, logdet):
    n = len(mu)
    D = np.array([O.distances(topo.parent, st.heights[b], st.rates[b], st.time_height[b], st.rate_mean[b]) for b in range(st.heights.shape[0])])
    dd = D - mu
    return -n * 0.9189385332046727 - 0.5 * (logdet + np.einsum("bi,bi->b", dd, (P @ dd.T).T))


def test_indefinite_precision_matrix_takes_the_product_form(gpu):
    """The reference evaluates dx . (P dx) with whatever precision matrix the record holds (app/Probability.hs:169, 183); the dense
    kernels need a factor and refuse a matrix without one (NotPositiveDefinite).  likelihood_function then evaluates the product form on
    the device (no factor needed): the value of the oracle's restatement of the reference's formula, for a Full and for a Sparse record."""
    topo = M.Topology(np.array([-1, 0, 1, 1, 0, 4, 4, 6, 6], dtype=np.int32))
    n = topo.n_nodes - 2
    rng = np.random.default_rng(5)
    A = rng.standard_normal((n, n))
    P = A + A.T                                              # symmetric, eigenvalues of both signs
    assert np.linalg.eigvalsh(P).min() < 0 < np.linalg.eigvalsh(P).max()
    mu = rng.uniform(0.05, 0.3, n)
    with pytest.raises(M.NotPositiveDefinite):
        M.MvnLikelihood(M.Full(mu, P, 1.7))
    s = M.State(1.0, 0.8, 1.3, np.array([1.0, 0.6, 0.0, 0.0, 0.7, 0.0, 0.3, 0.0, 0.0]), 0.9, 0.3,
                np.array([1.0, 1.1, 0.8, 1.2, 0.9, 1.05, 0.95, 1.3, 0.7]))
    d = O.distances(topo.parent, s.time_tree, s.rate_tree, s.time_height, s.rate_mean)
    want = O.logpdf_full(mu, P, 1.7, d)
    for lhd in (M.Full(mu, P, 1.7), M.Sparse(mu, [((i, j), float(P[i, j])) for i in range(n) for j in range(n)], 1.7)):
        f = M.likelihood_function(lhd, topo)
        assert abs(f(s) - want) <= 1e-12 * max(1.0, abs(want)), (f(s), want)
