"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Tolerances (fp64):
  real fixtures (cond <= 1e2):  |ll_gpu - ll_oracle| <= 1e-10 * max(1, |ll|)
  synthetic Sigma:              |ll_gpu - ll_oracle| <= 64 * N * eps * cond(Sigma) * max(1, q),  eps = 2^-53
  gradients:                    same bound per component relative to ||g||_inf
"""
import numpy as np
import pytest

import mcmc_date_amd as M
import oracle as O
from mcmc_date_amd import synthetic as S

pytestmark = pytest.mark.gpu

EPS = 2.0 ** -53


def rel_err(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


# ------------------------------------------------------------------------------------------
# golden fixtures (reference test inputs -> restated prepare -> oracle)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["06-leaves-constant-rate", "10-leaves-autocorrelated-rate",
                                  "12-leaves-variable-rate", "24-leaves-braces", "25-leaves-bastien"])
def test_fixture_rawx_host_and_device(gpu, golden, name):
    import torch

    fx = golden[name]
    lik = M.MvnLikelihood(M.Full(fx["mu"], fx["sigma_inv"], float(fx["logdet"])))
    ll_host = lik.logpdf(fx["X"])
    assert np.max(rel_err(ll_host, fx["ll_X"])) <= 1e-10
    Xd = torch.as_tensor(fx["X"], device=gpu)
    ll_dev = lik.logpdf(Xd).cpu().numpy()
    assert np.array_equal(ll_dev, ll_host)           # same kernel, same bits
    # single evaluation = drop-in for logDensityFullMultivariateNormal
    for i in (0, 5, len(fx["X"]) - 1):
        assert lik.logpdf1(fx["X"][i]) == ll_host[i]
    # the same operands given as the covariance matrix
    lik2 = M.MvnLikelihood.from_covariance(fx["mu"], fx["sigma"])
    assert abs(lik2.logdet_sigma - float(fx["logdet"])) <= 1e-10 * abs(float(fx["logdet"]))
    assert np.max(rel_err(lik2.logpdf(fx["X"]), fx["ll_X"])) <= 1e-10


@pytest.mark.parametrize("name", ["06-leaves-constant-rate", "10-leaves-autocorrelated-rate",
                                  "12-leaves-variable-rate", "24-leaves-braces", "25-leaves-bastien"])
def test_fixture_tree_states(gpu, golden, name):
    fx = golden[name]
    topo = M.Topology(fx["parent"])
    tl = M.MvnLikelihood(M.Full(fx["mu"], fx["sigma_inv"], float(fx["logdet"]))).bind_tree(topo)
    st = M.StateBatch(fx["H"], fx["R"], fx["tH"], fx["rMu"])
    ll, lj = tl.loglik(st)
    assert np.max(rel_err(ll, fx["ll_S"])) <= 1e-10
    assert np.max(rel_err(lj, fx["lj_S"])) <= 1e-12
    ll_d, lj_d = tl.loglik(st.to(gpu))
    assert np.array_equal(ll_d.cpu().numpy(), ll) and np.array_equal(lj_d.cpu().numpy(), lj)
    # the plugin closure: one state in, one log-likelihood out
    f = M.likelihood_function(M.Full(fx["mu"], fx["sigma_inv"], float(fx["logdet"])), topo)
    x = M.State(1.0, 1.0, float(fx["tH"][3]), fx["H"][3], float(fx["rMu"][3]), 1.0, fx["R"][3])
    assert f(x) == ll[3]


@pytest.mark.parametrize("name", ["12-leaves-variable-rate", "24-leaves-braces"])
def test_fixture_tree_gradient(gpu, golden, name):
    """Config 4: analytic gradient vs the oracle's and vs central finite differences of the oracle."""
    fx = golden[name]
    topo = M.Topology(fx["parent"])
    tl = M.MvnLikelihood(M.Full(fx["mu"], fx["sigma_inv"], float(fx["logdet"]))).bind_tree(topo)
    st = M.StateBatch(fx["H"], fx["R"], fx["tH"], fx["rMu"])
    ll, gH, gR, gt, gm = tl.grad(st)
    assert np.max(rel_err(ll, fx["ll_S"])) <= 1e-10
    ng = len(fx["gH"])
    scale = max(np.abs(fx["gH"]).max(), np.abs(fx["gR"]).max())
    assert np.max(np.abs(gH[:ng] - fx["gH"])) <= 1e-9 * scale
    assert np.max(np.abs(gR[:ng] - fx["gR"])) <= 1e-9 * scale
    assert np.max(rel_err(gt[:ng], fx["gtH"])) <= 1e-9 and np.max(rel_err(gm[:ng], fx["grMu"])) <= 1e-9
    # finite differences of the ORACLE log-likelihood (h = 1e-6 relative) for chain 1
    b = 1
    def ll_at(H, R, tH, rMu):
        return O.tree_loglik_full_batch(fx["parent"], H[None], R[None], np.array([tH]), np.array([rMu]), fx["mu"],
                                        fx["sigma_inv"], float(fx["logdet"]))[0][0]
    H0, R0 = fx["H"][b].copy(), fx["R"][b].copy()
    for v in (1, 2, topo.n_nodes - 2):
        for arr, g in ((H0, gH), (R0, gR)):
            if arr[v] == 0:
                continue
            h = 1e-6 * abs(arr[v])
            a = arr.copy(); a[v] += h
            c = arr.copy(); c[v] -= h
            if arr is H0:
                fd = (ll_at(a, R0, fx["tH"][b], fx["rMu"][b]) - ll_at(c, R0, fx["tH"][b], fx["rMu"][b])) / (2 * h)
            else:
                fd = (ll_at(H0, a, fx["tH"][b], fx["rMu"][b]) - ll_at(H0, c, fx["tH"][b], fx["rMu"][b])) / (2 * h)
            assert abs(fd - g[b, v]) <= 1e-5 * max(1.0, abs(fd)), (v, fd, g[b, v])
    # device-resident call returns the same bits
    out_d = tl.grad(st.to(gpu))
    assert np.array_equal(out_d[1].cpu().numpy(), gH) and np.array_equal(out_d[2].cpu().numpy(), gR)


def test_fixture_rawx_gradient(gpu, golden):
    fx = golden["24-leaves-braces"]
    lik = M.MvnLikelihood(M.Full(fx["mu"], fx["sigma_inv"], float(fx["logdet"])))
    ll, G = lik.grad(fx["X"])
    Gref = O.grad_full_batch(fx["mu"], fx["sigma_inv"], fx["X"])
    assert np.max(rel_err(ll, fx["ll_X"])) <= 1e-10
    assert np.max(np.abs(G - Gref)) <= 1e-10 * np.abs(Gref).max()


# ------------------------------------------------------------------------------------------
# the other LikelihoodData constructors (app/Probability.hs:210-235) run through the same kernel
# ------------------------------------------------------------------------------------------
def test_univariate_sparse_nodata(gpu, golden):
    fx = golden["12-leaves-variable-rate"]
    mu, X = fx["mu"], fx["X"]
    vs = np.diag(fx["sigma"]).copy()
    uni = M.MvnLikelihood(M.Univariate(mu, vs))
    ref = np.array([O.logpdf_univariate(mu, vs, x) for x in X])
    assert np.max(rel_err(uni.logpdf(X), ref)) <= 1e-10
    # sparse: drop small precision entries like toAssocMatrix (app/Main.hs:142-155) would
    P = fx["sigma_inv"].copy()
    P[np.abs(P) < 0.02 * np.abs(P).max()] = 0.0
    ii, jj = np.nonzero(P)
    assoc = [((int(i), int(j)), float(P[i, j])) for i, j in zip(ii, jj)]
    logdet = -np.linalg.slogdet(P)[1]
    sp = M.MvnLikelihood(M.Sparse(mu, assoc, logdet))
    ref = np.array([O.logpdf_sparse(mu, ii, jj, P[ii, jj], logdet, x) for x in X])
    assert np.max(rel_err(sp.logpdf(X), ref)) <= 1e-10
    nd = M.MvnLikelihood(M.NoData())
    assert np.all(nd.logpdf(X) == 0.0)                # likelihood 1.0, app/Probability.hs:281


# ------------------------------------------------------------------------------------------
# synthetic Sigma: every register-block size, ragged batches, strides
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,batch", [(1, 3), (2, 1), (9, 5), (63, 7), (64, 64), (65, 33), (128, 130), (129, 17),
                                     (200, 64), (256, 512), (257, 9), (384, 40), (500, 21), (768, 12), (1024, 16)])
def test_synthetic_logpdf_and_grad(gpu, n, batch):
    import torch

    mu, sigma = S.random_spd_problem(n, seed=n)
    X = S.sample_chains(mu, sigma, batch, seed=n)
    P = np.linalg.inv(sigma)
    logdet = np.linalg.slogdet(sigma)[1]
    kappa = np.linalg.cond(sigma)
    lik = M.MvnLikelihood.from_covariance(mu, sigma)
    ref = O.logpdf_full_batch(mu, P, logdet, X)
    q = -2.0 * (ref + 0.9189385332046727 * n) - logdet
    tol = 64 * n * EPS * kappa * np.maximum(1.0, q)
    ll = lik.logpdf(X)
    assert np.all(np.abs(ll - ref) <= tol), (np.max(np.abs(ll - ref)), tol.min())
    # device-resident, padded leading dimension
    ld = n + 3
    Xd = torch.zeros(batch, ld, dtype=torch.float64, device=gpu)
    Xd[:, :n] = torch.as_tensor(X, device=gpu)
    out = torch.empty(batch, dtype=torch.float64, device=gpu)
    M._capi.check(M._capi.lib().mcd_mvn_logpdf_batch(lik._h, Xd.data_ptr(), ld, batch, 1, None, out.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ll)
    ll2, G = lik.grad(X)
    split_ll = (n > 256 and batch <= 1024) or (240 < n <= 256 and batch <= 128)     # use_split (csrc/k_logpdf.hip)
    split_grad = batch <= 1024 and (n > 256 or (n > 240 and batch <= 512))          # use_split_grad
    if split_ll == split_grad:
        assert np.array_equal(ll2, ll)                       # the same forward product, the same bits
    else:
        assert np.all(np.abs(ll2 - ll) <= tol)               # row-split kernel (k_split.hip) on one side, sweeps on the other
    Gref = O.grad_full_batch(mu, P, X)
    gtol = 64 * n * EPS * kappa * np.abs(Gref).max() * 4
    assert np.max(np.abs(G - Gref)) <= gtol, (np.max(np.abs(G - Gref)), gtol)


@pytest.mark.parametrize("n,batch", [(193, 64), (200, 65), (224, 130), (255, 1000), (256, 512), (256, 1024), (241, 77), (256, 1), (230, 5), (199, 17), (129, 1), (160, 100), (192, 128)])
def test_row_split_form(gpu, n, batch, knobs):
    """128 < N <= 256 through the row-split form (k_split.hip; forced here with MCD_SPLIT=1 -- the automatic choice takes it at
    these sizes only up to 32 chains, see tests/test_gpu_split.py for N > 256): W's row blocks split over 8 workgroups per chain
    tile, partial sums added by the row group that started last.  Oracle bound of the sweeps, ragged tiles, padded rows,
    repeatable bits, the same bits from the host-pointer path (another stream, another scratch) and from concurrent callers."""
    import threading
    import torch

    knobs.setenv("MCD_SPLIT", "1")

    mu, sigma = S.random_spd_problem(n, seed=n)
    X = S.sample_chains(mu, sigma, batch, seed=n + 7)
    P = np.linalg.inv(sigma)
    logdet = np.linalg.slogdet(sigma)[1]
    kappa = np.linalg.cond(sigma)
    lik = M.MvnLikelihood.from_covariance(mu, sigma)
    ref = O.logpdf_full_batch(mu, P, logdet, X)
    q = -2.0 * (ref + 0.9189385332046727 * n) - logdet
    tol = 64 * n * EPS * kappa * np.maximum(1.0, q)
    ll = lik.logpdf(X)
    assert np.all(np.abs(ll - ref) <= tol), (np.max(np.abs(ll - ref)), tol.min())
    M.set_logpdf_form("sweep")
    try:
        sw = lik.logpdf(X)
    finally:
        M.set_logpdf_form("auto")
    assert np.all(np.abs(sw - ll) <= tol)
    assert batch < 64 or not np.array_equal(sw, ll)          # really another kernel (a few chains may agree to the last bit)
    ld = n + 3
    Xd = torch.full((batch, ld), np.nan, dtype=torch.float64, device=gpu)
    Xd[:, :n] = torch.as_tensor(X, device=gpu)
    out = torch.empty(batch, dtype=torch.float64, device=gpu)
    for _ in range(3):                                       # the tile counters are left at zero by every launch
        out.zero_()
        M._capi.check(M._capi.lib().mcd_mvn_logpdf_batch(lik._h, Xd.data_ptr(), ld, batch, 1, None, out.data_ptr()))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), ll)
    res = [None] * 4

    def work(k):
        res[k] = lik.logpdf(X)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(np.array_equal(r, ll) for r in res)
    if batch > 80:
        assert np.array_equal(lik.logpdf(X[:70]), ll[:70])   # a chain's value does not depend on the batch (same form)
    assert lik.logpdf1(X[0]) == ll[0]


@pytest.mark.parametrize("n,log10_kappa", [(45, 8), (64, 10), (256, 9), (300, 8)])
def test_ill_conditioned_precision_matrix(gpu, n, log10_kappa):
    """The reference's operand is Sigma^-1 (app/Main.hs:240) and real phylogenetic covariances are far worse conditioned than
    the fixtures (cond <= 1e2).  mcd_mvn_create factors Sigma^-1 = W^T W directly (no inverse of the matrix, no second
    factorisation): with cond(Sigma^-1) = 1e8 .. 1e10 every form still reproduces the reference's quadratic form
    dx^T Sigma^-1 dx, evaluated here in 80-bit arithmetic, to 1e-9 relative (an inverse-then-factor route loses cond^2 eps)."""
    rng = np.random.default_rng(n)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.logspace(0.0, float(log10_kappa), n)
    P = (Q * lam) @ Q.T
    P = 0.5 * (P + P.T)
    logdet = -float(np.sum(np.log(lam)))                 # log det Sigma
    mu = rng.uniform(0.01, 1.0, n)
    X = mu + rng.standard_normal((40, n)) * 1e-3
    Pl, dxl = P.astype(np.longdouble), (X - mu).astype(np.longdouble)
    q_ref = np.einsum("bi,ij,bj->b", dxl, Pl, dxl)
    ll_ref = (-0.9189385332046727 * n - 0.5 * (np.longdouble(logdet) + q_ref)).astype(np.float64)
    lik = M.MvnLikelihood(M.Full(mu, P, logdet))
    for form in ("sweep", "multiply", "auto"):
        lik.set_form(form)
        ll = lik.logpdf(X)
        q = -2.0 * (ll + 0.9189385332046727 * n) - logdet
        assert np.max(np.abs(q - q_ref.astype(np.float64)) / q_ref.astype(np.float64)) <= 1e-9, form
        assert np.max(np.abs(ll - ll_ref) / np.abs(ll_ref)) <= 1e-9, form
    lik.set_form("auto")
    # the oracle in the reference's own fp64 algebra agrees with the 80-bit value no better than that
    ref64 = O.logpdf_full_batch(mu, P, logdet, X)
    assert np.max(np.abs(ref64 - ll_ref) / np.abs(ll_ref)) <= 1e-9
    # gradient: -Sigma^-1 dx
    _, G = lik.grad(X[:8])
    Gref = -(dxl[:8] @ Pl).astype(np.float64)
    assert np.max(np.abs(G - Gref)) <= 1e-8 * np.abs(Gref).max()


def test_large_batch_two_chains_per_wave(gpu):
    """Batches above 8192 chains switch to two chains per wave; same results per chain."""
    import torch

    n, batch = 96, 8192 + 1027
    mu, sigma = S.random_spd_problem(n, seed=5)
    X = S.sample_chains(mu, sigma, batch, seed=5)
    lik = M.MvnLikelihood.from_covariance(mu, sigma)
    M.set_logpdf_form("sweep")
    try:
        ll = lik.logpdf(torch.as_tensor(X, device=gpu)).cpu().numpy()
        small = lik.logpdf(X[:300])
    finally:
        M.set_logpdf_form("auto")
    assert np.array_equal(ll[:300], small)
    ref = O.logpdf_full_batch(mu, np.linalg.inv(sigma), np.linalg.slogdet(sigma)[1], X[-50:])
    assert np.max(rel_err(ll[-50:], ref)) <= 1e-9


@pytest.mark.parametrize("n", [100, 200, 300, 500])
def test_every_launch_geometry(gpu, n):
    """The three workgroup geometries (<= 512, <= 4096, > 4096 chains) must agree chain by chain.
    Regression: a noalias pointer into the LDS ring once let the compiler reuse a slot's values across
    the barrier in some instantiations only."""
    mu, sigma = S.random_spd_problem(n, seed=n)
    P = np.linalg.inv(sigma)
    logdet = np.linalg.slogdet(sigma)[1]
    lik = M.MvnLikelihood.from_covariance(mu, sigma)
    X = S.sample_chains(mu, sigma, 5000, seed=n)
    ref = O.logpdf_full_batch(mu, P, logdet, X[:40])
    M.set_logpdf_form("sweep")                       # (large batches would otherwise take the multiply form)
    try:
        outs = [lik.logpdf(X[:B])[:40] for B in (40, 600, 5000)]
        ll, G = lik.grad(X[:600])
    finally:
        M.set_logpdf_form("auto")
    for o in outs:
        assert np.max(rel_err(o, ref)) <= 1e-9
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    Gref = O.grad_full_batch(mu, P, X[:8])
    assert np.max(np.abs(G[:8] - Gref)) <= 1e-9 * np.abs(Gref).max()
    assert np.array_equal(ll[:40], outs[0])


def test_synthetic_tree_256(gpu):
    """Config 3 tree variant: L = 129 leaves -> N = 255 (2L-3 is odd; 256 itself cannot be a tree)."""
    topo = S.random_topology(129, seed=256)
    n = topo.n_nodes - 2
    mu, sigma = S.random_spd_problem(n, seed=256)
    st = S.random_states(topo, 96, seed=256)
    # put the states near mu: choose rMu so that distances have the scale of mu
    P = np.linalg.inv(sigma)
    logdet = np.linalg.slogdet(sigma)[1]
    tl = M.MvnLikelihood(M.Full(mu, P, logdet)).bind_tree(topo)
    ll, lj = tl.loglik(st)
    ref, refj = O.tree_loglik_full_batch(topo.parent, st.heights, st.rates, st.time_height, st.rate_mean, mu, P, logdet)
    assert np.max(np.abs(ll - ref) / np.abs(ref)) <= 1e-11
    assert np.max(rel_err(lj, refj)) <= 1e-12
    out = tl.grad(st)
    for b in (0, 17, 95):
        gH, gR, gt, gm = O.tree_grad_full(topo.parent, st.heights[b], st.rates[b], st.time_height[b], st.rate_mean[b], mu, P)
        sc = max(np.abs(gH).max(), np.abs(gR).max())
        assert np.max(np.abs(out[1][b] - gH)) <= 1e-9 * sc and np.max(np.abs(out[2][b] - gR)) <= 1e-9 * sc
        assert abs(out[3][b] - gt) <= 1e-9 * abs(gt) and abs(out[4][b] - gm) <= 1e-9 * abs(gm)


def test_tree_large_batch_two_chains_per_wave(gpu):
    """Tree states: batches above 4096 chains use two chains per compute wave; per chain the same bits as a small batch,
    and the oracle's values."""
    topo = S.random_topology(50, seed=7)
    n = topo.n_nodes - 2
    mu, sigma = S.random_spd_problem(n, seed=7)
    P = np.linalg.inv(sigma)
    logdet = np.linalg.slogdet(sigma)[1]
    batch = 4096 + 1500 + 3
    st = S.random_states(topo, batch, seed=9)
    tl = M.MvnLikelihood(M.Full(mu, P, logdet)).bind_tree(topo)
    M.set_logpdf_form("sweep")
    try:
        ll, lj = tl.loglik(st)
        ll_s, lj_s = tl.loglik(st.slice(batch - 260, batch))
    finally:
        M.set_logpdf_form("auto")
    assert np.array_equal(ll[-260:], ll_s) and np.array_equal(lj[-260:], lj_s)
    ref, refj = O.tree_loglik_full_batch(topo.parent, st.heights[-40:], st.rates[-40:], st.time_height[-40:], st.rate_mean[-40:], mu, P, logdet)
    assert np.max(np.abs(ll[-40:] - ref) / np.abs(ref)) <= 1e-11 and np.max(rel_err(lj[-40:], refj)) <= 1e-12


# ------------------------------------------------------------------------------------------
# multiply form (k_wide.hip): z = L^-1 (x - mu) on the fp64 matrix cores, used for large batches
# ------------------------------------------------------------------------------------------
@pytest.fixture
def multiply_form(gpu):
    prev = M.set_logpdf_form("multiply")
    yield
    M.set_logpdf_form(prev)


@pytest.mark.parametrize("n,batch", [(1, 3), (2, 1), (9, 5), (15, 16), (16, 17), (17, 100), (63, 7), (64, 64), (65, 33), (129, 17), (200, 64),
                                     (255, 1500), (256, 512), (257, 9), (272, 40), (500, 21), (513, 70), (768, 12), (1024, 16)])
def test_multiply_form_logpdf(gpu, multiply_form, n, batch):
    """Every shape of the tiling: N below one row block, partial and several super blocks (N > 256 walks the chunks
    0 .. s), ragged chain tiles, padded leading dimension.  Same bound as the column sweep."""
    import torch

    mu, sigma = S.random_spd_problem(n, seed=n)
    X = S.sample_chains(mu, sigma, batch, seed=n)
    P = np.linalg.inv(sigma)
    logdet = np.linalg.slogdet(sigma)[1]
    kappa = np.linalg.cond(sigma)
    lik = M.MvnLikelihood.from_covariance(mu, sigma)
    ref = O.logpdf_full_batch(mu, P, logdet, X)
    q = -2.0 * (ref + 0.9189385332046727 * n) - logdet
    tol = 64 * n * EPS * kappa * np.maximum(1.0, q)
    ll = lik.logpdf(X)
    assert np.all(np.abs(ll - ref) <= tol), (np.max(np.abs(ll - ref)), tol.min())
    ld = n + 3
    Xd = torch.full((batch, ld), np.nan, dtype=torch.float64, device=gpu)       # the padding must never be read
    Xd[:, :n] = torch.as_tensor(X, device=gpu)
    out = torch.empty(batch, dtype=torch.float64, device=gpu)
    M._capi.check(M._capi.lib().mcd_mvn_logpdf_batch(lik._h, Xd.data_ptr(), ld, batch, 1, None, out.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ll)
    # a chain's value does not depend on its neighbours or on the batch it travels in
    assert lik.logpdf1(X[batch - 1]) == ll[batch - 1]
    M.set_logpdf_form("sweep")
    sw = lik.logpdf(X)
    M.set_logpdf_form("multiply")
    assert np.all(np.abs(sw - ll) <= tol)


@pytest.mark.parametrize("ct", [4100, 17000])
def test_multiply_form_chain_tiles(gpu, multiply_form, ct):
    """16, 32 and 64 chains per workgroup (chosen by batch size) give each chain the same bits."""
    n = 130
    mu, sigma = S.random_spd_problem(n, seed=3)
    X = S.sample_chains(mu, sigma, ct, seed=3)
    lik = M.MvnLikelihood.from_covariance(mu, sigma)
    big = lik.logpdf(X)
    small = lik.logpdf(X[:200])
    assert np.array_equal(big[:200], small)
    tail = lik.logpdf(X[-77:])
    assert np.array_equal(big[-77:], tail)
    ref = O.logpdf_full_batch(mu, np.linalg.inv(sigma), np.linalg.slogdet(sigma)[1], X[-50:])
    assert np.max(rel_err(big[-50:], ref)) <= 1e-9


@pytest.mark.parametrize("leaves,batch", [(3, 5), (12, 40), (50, 333), (129, 96), (140, 50), (400, 20)])
def test_multiply_form_tree(gpu, multiply_form, leaves, batch):
    """Tree states through the multiply form: oracle values, and the root-branch Jacobian bit for bit as the sweep's."""
    topo = S.random_topology(leaves, seed=leaves)
    n = topo.n_nodes - 2
    mu, sigma = S.random_spd_problem(n, seed=leaves)
    P = np.linalg.inv(sigma)
    logdet = np.linalg.slogdet(sigma)[1]
    st = S.random_states(topo, batch, seed=leaves)
    tl = M.MvnLikelihood(M.Full(mu, P, logdet)).bind_tree(topo)
    ll, lj = tl.loglik(st)
    ref, refj = O.tree_loglik_full_batch(topo.parent, st.heights, st.rates, st.time_height, st.rate_mean, mu, P, logdet)
    assert np.max(np.abs(ll - ref) / np.abs(ref)) <= 1e-11
    assert np.max(rel_err(lj, refj)) <= 1e-12
    ll_d, lj_d = tl.loglik(st.to(gpu))
    assert np.array_equal(ll_d.cpu().numpy(), ll) and np.array_equal(lj_d.cpu().numpy(), lj)
    ll_n, none = tl.loglik(st, want_jacobian=False)
    assert none is None and np.array_equal(ll_n, ll)
    M.set_logpdf_form("sweep")
    ll_s, lj_s = tl.loglik(st)
    M.set_logpdf_form("multiply")
    assert np.array_equal(lj_s, lj)
    assert np.max(np.abs(ll_s - ll) / np.abs(ll)) <= 1e-12


@pytest.mark.parametrize("n,batch", [(1, 3), (9, 5), (16, 17), (17, 100), (63, 7), (64, 64), (65, 33), (129, 17), (200, 64), (255, 1500), (256, 520),
                                     (257, 9), (272, 40), (500, 21), (513, 70), (768, 4200), (1024, 16)])
def test_multiply_form_gradient(gpu, multiply_form, n, batch):
    """ll and d ll / d x through the two triangular MFMA products, against the oracle and the sweeps: N <= 256 with z and y
    in one LDS chunk, above through the gradient buffer itself (several chunks, partial super blocks, ragged tiles)."""
    import torch

    mu, sigma = S.random_spd_problem(n, seed=n)
    X = S.sample_chains(mu, sigma, batch, seed=n + 1)
    P = np.linalg.inv(sigma)
    logdet = np.linalg.slogdet(sigma)[1]
    kappa = np.linalg.cond(sigma)
    lik = M.MvnLikelihood.from_covariance(mu, sigma)
    ll, G = lik.grad(X)
    assert np.array_equal(ll, lik.logpdf(X))                 # same forward product, same bits
    Gref = O.grad_full_batch(mu, P, X)
    gtol = 64 * n * EPS * kappa * np.abs(Gref).max() * 4
    assert np.max(np.abs(G - Gref)) <= gtol, (np.max(np.abs(G - Gref)), gtol)
    ll_d, G_d = lik.grad(torch.as_tensor(X, device=gpu))
    assert np.array_equal(ll_d.cpu().numpy(), ll) and np.array_equal(G_d.cpu().numpy(), G)
    # padded gradient rows: the padding is never written (the rows double as scratch for z above 256)
    ldg = n + 5
    Gp = torch.full((batch, ldg), -7.0, dtype=torch.float64, device=gpu)
    llp = torch.empty(batch, dtype=torch.float64, device=gpu)
    Xd = torch.as_tensor(X, device=gpu)
    M._capi.check(M._capi.lib().mcd_mvn_grad_batch(lik._h, Xd.data_ptr(), n, batch, 1, None, llp.data_ptr(), Gp.data_ptr(), ldg))
    torch.cuda.synchronize()
    assert np.array_equal(Gp[:, :n].cpu().numpy(), G) and bool((Gp[:, n:] == -7.0).all())
    M.set_logpdf_form("sweep")
    ll_s, G_s = lik.grad(X)
    M.set_logpdf_form("multiply")
    assert np.max(np.abs(G_s - G)) <= gtol and np.max(rel_err(ll_s, ll)) <= 1e-12
    if n > 1:
        _, G0 = lik.grad(mu[None, :])
        assert np.all(G0 == 0.0)


@pytest.mark.parametrize("leaves,batch", [(3, 5), (4, 33), (12, 40), (50, 333), (100, 70), (129, 96), (130, 37), (140, 50), (300, 2100), (513, 20)])
def test_multiply_form_tree_gradient(gpu, multiply_form, leaves, batch):
    """Chain rule to heights, rates, tH, rMu on the LDS copy of g (N <= 256) or in sub-batches of chains with g parked in the
    height-gradient rows (above): oracle values per chain, sweep values for all."""
    topo = S.random_topology(leaves, seed=leaves)
    n = topo.n_nodes - 2
    mu, sigma = S.random_spd_problem(n, seed=leaves)
    P = np.linalg.inv(sigma)
    logdet = np.linalg.slogdet(sigma)[1]
    st = S.random_states(topo, batch, seed=leaves + 3)
    tl = M.MvnLikelihood(M.Full(mu, P, logdet)).bind_tree(topo)
    out = tl.grad(st)
    ll, _ = tl.loglik(st)
    assert np.array_equal(out[0], ll)
    for b in sorted({0, batch // 2, batch - 1}):
        gH, gR, gt, gm = O.tree_grad_full(topo.parent, st.heights[b], st.rates[b], st.time_height[b], st.rate_mean[b], mu, P)
        sc = max(np.abs(gH).max(), np.abs(gR).max())
        assert np.max(np.abs(out[1][b] - gH)) <= 1e-9 * sc and np.max(np.abs(out[2][b] - gR)) <= 1e-9 * sc
        assert abs(out[3][b] - gt) <= 1e-9 * abs(gt) and abs(out[4][b] - gm) <= 1e-9 * abs(gm)
    out_d = tl.grad(st.to(gpu))
    for a, b in zip(out, out_d):
        assert np.array_equal(a, b.cpu().numpy())
    M.set_logpdf_form("sweep")
    sw = tl.grad(st)
    M.set_logpdf_form("multiply")
    sc = max(np.abs(sw[1]).max(), np.abs(sw[2]).max())
    assert np.max(np.abs(sw[1] - out[1])) <= 1e-10 * sc and np.max(np.abs(sw[2] - out[2])) <= 1e-10 * sc
    assert np.max(np.abs(sw[3] - out[3]) / np.abs(sw[3])) <= 1e-9 and np.max(np.abs(sw[4] - out[4]) / np.abs(sw[4])) <= 1e-9


def test_multiply_form_random_shapes(gpu, multiply_form):
    """Seeded random dimensions and batch sizes (partial row blocks, partial super blocks, ragged chain tiles) for ll and
    gradient, raw x and tree states, against the column sweeps chain by chain."""
    rng = np.random.default_rng(20251003)
    for case in range(36):
        n = int(rng.integers(1, 1025)) if case % 3 else int(rng.choice([15, 16, 17, 31, 240, 241, 255, 256, 257, 271, 272, 273, 511, 512, 513, 1023, 1024]))
        batch = int(rng.integers(1, 200))
        mu, sigma = S.random_spd_problem(n, seed=1000 + case)
        X = S.sample_chains(mu, sigma, batch, seed=case)
        lik = M.MvnLikelihood.from_covariance(mu, sigma)
        ll, G = lik.grad(X)
        ll1 = lik.logpdf(X)
        M.set_logpdf_form("sweep")
        ll_s, G_s = lik.grad(X)
        M.set_logpdf_form("multiply")
        kappa = np.linalg.cond(sigma)
        assert np.array_equal(ll, ll1), (n, batch)
        assert np.max(rel_err(ll, ll_s)) <= 64 * n * EPS * kappa, (n, batch, np.max(rel_err(ll, ll_s)))
        assert np.max(np.abs(G - G_s)) <= 256 * n * EPS * kappa * np.abs(G_s).max(), (n, batch)
    for case in range(14):
        leaves = int(rng.integers(3, 514))
        batch = int(rng.integers(1, 120))
        topo = S.random_topology(leaves, seed=case)
        n = topo.n_nodes - 2
        mu, sigma = S.random_spd_problem(n, seed=2000 + case)
        st = S.random_states(topo, batch, seed=case)
        tl = M.MvnLikelihood.from_covariance(mu, sigma).bind_tree(topo)
        out = tl.grad(st)
        ll1, lj1 = tl.loglik(st)
        M.set_logpdf_form("sweep")
        sw = tl.grad(st)
        ll_s, lj_s = tl.loglik(st)
        M.set_logpdf_form("multiply")
        assert np.array_equal(out[0], ll1) and np.array_equal(lj1, lj_s), (leaves, batch)
        assert np.max(np.abs(ll1 - ll_s) / np.abs(ll_s)) <= 1e-11, (leaves, batch)
        sc = max(np.abs(sw[1]).max(), np.abs(sw[2]).max())
        assert np.max(np.abs(sw[1] - out[1])) <= 1e-9 * sc and np.max(np.abs(sw[2] - out[2])) <= 1e-9 * sc, (leaves, batch)
        assert np.max(np.abs(sw[3] - out[3]) / np.abs(sw[3])) <= 1e-8 and np.max(np.abs(sw[4] - out[4]) / np.abs(sw[4])) <= 1e-8, (leaves, batch)


def test_in_place_gradient(gpu, multiply_form):
    """G may be the very array that holds x (a chain's x is read before its gradient is written): the multiply form up to
    N = 256 allows it, above that the call falls back to the sweep, which does as well."""
    import torch

    for n, batch in ((200, 300), (300, 70)):
        mu, sigma = S.random_spd_problem(n, seed=n)
        X = S.sample_chains(mu, sigma, batch, seed=n)
        lik = M.MvnLikelihood.from_covariance(mu, sigma)
        ll, G = lik.grad(X)
        Xd = torch.as_tensor(X, device=gpu).clone()
        out = torch.empty(batch, dtype=torch.float64, device=gpu)
        M._capi.check(M._capi.lib().mcd_mvn_grad_batch(lik._h, Xd.data_ptr(), n, batch, 1, None, out.data_ptr(), Xd.data_ptr(), n))
        torch.cuda.synchronize()
        assert np.max(rel_err(out.cpu().numpy(), ll)) <= 1e-12
        assert np.max(np.abs(Xd.cpu().numpy() - G)) <= 1e-10 * np.abs(G).max()


def test_multiply_form_first_launch_under_capture(gpu):
    """The large dynamic LDS of the multiply-form kernels is allowed when the handle is created, so the very first launch
    may already sit inside a stream capture (a sampler that replays its step from a hipGraph)."""
    import torch

    n, batch = 144, 4100
    mu, sigma = S.random_spd_problem(n, seed=21)
    lik = M.MvnLikelihood.from_covariance(mu, sigma)
    X = torch.as_tensor(S.sample_chains(mu, sigma, batch, seed=21), device=gpu)
    out = torch.zeros(batch, dtype=torch.float64, device=gpu)
    G = torch.zeros_like(X)
    ll2 = torch.zeros_like(out)
    L = M._capi.lib()
    stream = torch.cuda.Stream(device=gpu)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        lik.logpdf_into(X, out)
        M._capi.check(L.mcd_mvn_grad_batch(lik._h, X.data_ptr(), X.stride(0), batch, 1, torch.cuda.current_stream().cuda_stream,
                                           ll2.data_ptr(), G.data_ptr(), G.stride(0)))
    g.replay()
    torch.cuda.synchronize()
    ref_ll, ref_G = lik.grad(X)
    assert torch.equal(out, ref_ll) and torch.equal(ll2, ref_ll) and torch.equal(G, ref_G)
    assert bool(torch.isfinite(out).all()) and float(out.abs().max()) > 0.0


def test_form_selection(gpu):
    """auto = sweep for a sampler's usual batch, multiply for thousands of chains of a large tree; unknown values are
    refused; non-finite inputs flow through the multiply form as through the sweep."""
    n = 160
    mu, sigma = S.random_spd_problem(n, seed=11)
    lik = M.MvnLikelihood.from_covariance(mu, sigma)
    X = S.sample_chains(mu, sigma, 3000, seed=11)
    assert M.set_logpdf_form("auto") == "auto"
    auto_small, auto_big = lik.logpdf(X[:256]), lik.logpdf(X)
    M.set_logpdf_form("sweep")
    sw_small, sw_big = lik.logpdf(X[:256]), lik.logpdf(X)
    assert M.set_logpdf_form("multiply") == "sweep"
    mu_big = lik.logpdf(X)
    Xn = X[:40].copy()
    Xn[3, 7] = np.nan
    Xn[5, 0] = np.inf
    bad = lik.logpdf(Xn)
    _, G_mu = lik.grad(X)
    assert M.set_logpdf_form("auto") == "multiply"
    _, G_auto = lik.grad(X)
    assert np.array_equal(G_auto, G_mu)                      # gradients follow the same choice (N <= 256)
    assert np.array_equal(auto_small, sw_small) and np.array_equal(auto_big, mu_big)
    assert not np.array_equal(auto_big, sw_big) and np.max(rel_err(auto_big, sw_big)) <= 1e-12
    assert np.isnan(bad[3]) and (np.isnan(bad[5]) or bad[5] == -np.inf) and np.all(np.isfinite(np.delete(bad, [3, 5])))
    with pytest.raises(ValueError):
        M.set_logpdf_form("fast")
    assert M._capi.lib().mcd_set_logpdf_form(7) == M._capi.MCD_ERR_INVALID_ARG
    assert M.set_logpdf_form("auto") == "auto"


# ------------------------------------------------------------------------------------------
# size-independent properties at BASELINE sizes
# ------------------------------------------------------------------------------------------
def test_properties_full_size(gpu):
    import torch

    n, batch = 256, 512
    mu, sigma = S.random_spd_problem(n, seed=256)
    lik = M.MvnLikelihood.from_covariance(mu, sigma)
    L = lik.cholesky_factor()
    assert np.allclose(L @ L.T, sigma, rtol=1e-12, atol=1e-18)
    rng = np.random.default_rng(1)
    Z = rng.standard_normal((batch, n))
    X = mu + Z @ L.T                                   # then L^-1 (x - mu) = z exactly up to rounding
    ll = lik.logpdf(torch.as_tensor(X, device=gpu)).cpu().numpy()
    q_expected = np.sum(Z * Z, axis=1)
    c = -0.9189385332046727 * n
    q = -2.0 * (ll - c) - lik.logdet_sigma
    assert np.max(np.abs(q - q_expected) / q_expected) <= 1e-10
    # maximum at mu, symmetry around mu, determinism
    ll_mu = lik.logpdf1(mu)
    assert ll_mu == c - 0.5 * lik.logdet_sigma and np.all(ll <= ll_mu)
    ll_ref = lik.logpdf(2 * mu[None, :] - X)
    assert np.max(np.abs(ll_ref - ll) / np.abs(ll)) <= 1e-12
    assert np.array_equal(lik.logpdf(X), ll)
    # gradient is linear in (x - mu):  G(mu + a d) = a G(mu + d);  G(mu) = 0
    _, G1 = lik.grad(X[:8])
    _, G2 = lik.grad(mu + 2.0 * (X[:8] - mu))
    assert np.max(np.abs(G2 - 2.0 * G1)) <= 1e-9 * np.abs(G1).max()
    _, G0 = lik.grad(mu[None, :])
    assert np.all(G0 == 0.0)


def test_properties_full_size_many_chains(gpu):
    """BASELINE's N = 256 at 8192 chains (the automatic choice is the multiply form there): the quadratic form of
    x = mu + L z is |z|^2, the maximum sits at mu exactly, the density is symmetric around mu, runs repeat bit for bit,
    and a chain's value is the same in a batch of 8192 (32 chains per workgroup) and of 2100 (16 per workgroup)."""
    import torch

    n, batch = 256, 8192
    mu, sigma = S.random_spd_problem(n, seed=256)
    lik = M.MvnLikelihood.from_covariance(mu, sigma)
    L = lik.cholesky_factor()
    rng = np.random.default_rng(2)
    Z = rng.standard_normal((batch, n))
    X = mu + Z @ L.T
    X[17] = mu
    Xd = torch.as_tensor(X, device=gpu)
    ll = lik.logpdf(Xd).cpu().numpy()
    c = -0.9189385332046727 * n
    q = -2.0 * (ll - c) - lik.logdet_sigma
    q_expected = np.sum(Z * Z, axis=1)
    keep = np.arange(batch) != 17
    assert np.max(np.abs(q[keep] - q_expected[keep]) / q_expected[keep]) <= 1e-10
    assert ll[17] == c - 0.5 * lik.logdet_sigma and np.all(ll <= ll[17])
    ll_ref = lik.logpdf(torch.as_tensor(2 * mu[None, :] - X, device=gpu)).cpu().numpy()
    assert np.max(np.abs(ll_ref - ll) / np.abs(ll)) <= 1e-12
    assert np.array_equal(lik.logpdf(Xd).cpu().numpy(), ll)
    assert np.array_equal(lik.logpdf(X[:2100]), ll[:2100])
    M.set_logpdf_form("sweep")
    try:
        sw = lik.logpdf(Xd).cpu().numpy()
    finally:
        M.set_logpdf_form("auto")
    assert not np.array_equal(sw, ll) and np.max(np.abs(sw - ll) / np.abs(ll)) <= 1e-13


# ------------------------------------------------------------------------------------------
# edge cases and error behaviour
# ------------------------------------------------------------------------------------------
def test_nonfinite_inputs_flow_through(gpu, golden):
    """NaN/Inf in a proposal must come out as NaN / -Inf (never finite, never an error): the sampler
    rejects such proposals (lib/Mcmc/Tree/Proposal/Unconstrained.hs:304-306)."""
    fx = golden["24-leaves-braces"]
    lik = M.MvnLikelihood(M.Full(fx["mu"], fx["sigma_inv"], float(fx["logdet"])))
    X = fx["X"][:6].copy()
    X[1, 3] = np.nan
    X[2, 0] = np.inf
    X[3, -1] = -np.inf
    X[4, 7] = 1e308
    ll = lik.logpdf(X)
    assert np.isfinite(ll[0]) and np.isfinite(ll[5])
    assert np.isnan(ll[1])
    for b in (2, 3, 4):
        assert not np.isfinite(ll[b]) and not (ll[b] > 0)
        ref = O.logpdf_full(fx["mu"], fx["sigma_inv"], float(fx["logdet"]), X[b])
        assert not np.isfinite(ref)
    # untouched neighbours are bit-identical to a clean run
    assert ll[0] == lik.logpdf(fx["X"][:1])[0]


def test_empty_batch_and_errors(gpu, golden):
    fx = golden["06-leaves-constant-rate"]
    lik = M.MvnLikelihood(M.Full(fx["mu"], fx["sigma_inv"], float(fx["logdet"])))
    assert lik.logpdf(np.zeros((0, 9))).shape == (0,)
    with pytest.raises(ValueError):
        lik.logpdf(np.zeros((2, 8)))
    bad = fx["sigma"].copy()
    bad[0, 0] = -1.0
    with pytest.raises(M.NotPositiveDefinite):
        M.MvnLikelihood.from_covariance(fx["mu"], bad)
    with pytest.raises(M.McdError):
        M.MvnLikelihood.from_covariance(np.zeros(1025), np.eye(1025))
    # non-bifurcating root: app/Tools.hs:43
    tri = M.Topology(np.array([-1, 0, 0, 0, 3, 3, 3, 3, 3, 3, 3], np.int32))
    with pytest.raises(M.RootNotBifurcating):
        lik.bind_tree(tri)
    # wrong number of branches
    with pytest.raises(M.McdError):
        lik.bind_tree(M.Topology(np.array([-1, 0, 0], np.int32)))


def test_concurrent_callers(gpu, golden):
    """The closure is called from several OS threads in the reference (-threaded -N, Parallel)."""
    import threading

    fx = golden["24-leaves-braces"]
    lik = M.MvnLikelihood(M.Full(fx["mu"], fx["sigma_inv"], float(fx["logdet"])))
    expect = lik.logpdf(fx["X"])
    errs = []

    def work(k):
        for _ in range(20):
            out = lik.logpdf(fx["X"][k::4])
            if not np.array_equal(out, expect[k::4]):
                errs.append(k)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs


def test_cpp_host_mirror(gpu, golden, tmp_path):
    """The C++ mirror of the reference's plugin surface (mcmc-date_amd/host/mcmcdate.hpp) through the C ABI."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "test_host_mirror")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    fx = golden["12-leaves-variable-rate"]
    b = 5
    p = tmp_path / "fixture.txt"
    with open(p, "w") as f:
        f.write(f"{len(fx['parent'])}\n" + " ".join(map(str, fx["parent"])) + "\n")
        f.write(" ".join(repr(float(v)) for v in fx["mu"]) + "\n")
        f.write(" ".join(repr(float(v)) for v in fx["sigma_inv"].ravel()) + "\n")
        f.write(repr(float(fx["logdet"])) + "\n")
        f.write(f"{float(fx['tH'][b])!r} {float(fx['rMu'][b])!r}\n")
        f.write(" ".join(repr(float(v)) for v in fx["H"][b]) + "\n")
        f.write(" ".join(repr(float(v)) for v in fx["R"][b]) + "\n")
        f.write(f"{float(fx['ll_S'][b])!r} {float(fx['lj_S'][b])!r}\n")
    r = subprocess.run([exe, str(p)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("ok")
