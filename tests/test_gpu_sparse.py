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
    """A symmetric, strictly diagonally dominant (hence SPD) precision matrix: a band plus `extra` random off-diagonal entries per
    row, values of mixed sign; returns (scipy CSR, association list)."""
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    for d in range(1, band + 1):
        i = np.arange(n - d)
        v = rng.uniform(-1.0, 1.0, n - d)
        rows += [i, i + d]; cols += [i + d, i]; vals += [v, v]
    i = rng.integers(0, n, n * extra)
    j = rng.integers(0, n, n * extra)
    keep = np.abs(i - j) > band
    i, j = i[keep], j[keep]
    v = rng.uniform(-0.5, 0.5, len(i))
    rows += [i, j]; cols += [j, i]; vals += [v, v]
    A = sps.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
    A.sum_duplicates()
    diag = np.abs(A).sum(axis=1).A1 + rng.uniform(0.5, 1.5, n)
    P = (A + sps.diags(diag)).tocsr()
    P = P * 1e3                                              # precisions of branch lengths ~ 1e-2: like the fixtures
    coo = P.tocoo()
    assoc = [((int(a), int(b)), float(c)) for a, b, c in zip(coo.row, coo.col, coo.data)]
    return P, assoc


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
    assert np.array_equal(ll_g, ll)
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
    assert np.array_equal(ll2.cpu().numpy(), ll)
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
