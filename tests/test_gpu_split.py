"""GPU tests of the row-split multiply form (csrc/k_split.hip): 128 < N <= 1024, 1 .. 1024 chains, raw x and tree states.

* parity with the oracle (bound of tests/test_gpu_parity.py) and with the column sweep over every staging shape (1 .. 4 chunks of
  256 columns), every number of row groups (G = 8 / 16 / 32), ragged tiles, padded rows;
* both gradients on the same schedule (two triangular products, the second over the reversed rows; chain rule for tree states);
* BASELINE config 5's per-GPU share at size: N = 1024 x 512 chains (raw x, 1023-node tree, both gradients);
* the cross-workgroup hand-over under stress: > 10^6 back-to-back launches with ALTERNATING inputs, every result compared, a
  memory-heavy kernel running beside them, the row groups of a tile on one XCD and spread over all eight;
* scratch ownership: a graph captured on one stream replayed on another while eager launches go on on the capture stream.
"""
import ctypes as C
import os

import numpy as np
import pytest

import mcmc_date_amd as M
import oracle as O
from mcmc_date_amd import synthetic as S

pytestmark = pytest.mark.gpu

EPS = 2.0 ** -53
LN_SQRT_2PI = 0.9189385332046727


class env:
    """Set knobs of the launcher (mcd_set_option: MCD_SPLIT, MCD_SPLIT_G, MCD_SPLIT_SCATTER) for a block -- until round 3 these were
    environment variables read per launch."""

    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        import mcmc_date_amd as M

        self.old = {k: M.get_option(k) for k in self.kv}
        for k, v in self.kv.items():
            M.set_option(k, v)

    def __exit__(self, *a):
        import mcmc_date_amd as M

        for k, v in self.old.items():
            M.set_option(k, v)


@pytest.fixture(autouse=True)
def split_wherever_possible():
    """The automatic choice takes the row split above N = 256 (240 < N <= 256: up to 128 chains); these tests exercise it over
    its whole range (N > 128, up to 1024 chains): the knob MCD_SPLIT = 1."""
    with env(MCD_SPLIT=1):
        yield


def problem(n, batch, seed):
    mu, sigma = S.random_spd_problem(n, seed=seed)
    X = S.sample_chains(mu, sigma, batch, seed=seed + 7)
    P = np.linalg.inv(sigma)
    logdet = np.linalg.slogdet(sigma)[1]
    kappa = np.linalg.cond(sigma)
    lik = M.MvnLikelihood.from_covariance(mu, sigma)
    ref = O.logpdf_full_batch(mu, P, logdet, X)
    q = -2.0 * (ref + LN_SQRT_2PI * n) - logdet
    tol = 64 * n * EPS * kappa * np.maximum(1.0, q)
    return mu, sigma, P, logdet, X, lik, ref, tol


@pytest.mark.parametrize("n,batch", [(257, 9), (300, 64), (384, 512), (500, 21), (512, 1024), (513, 100), (700, 33), (768, 512),
                                     (769, 17), (1000, 130), (1023, 1), (1024, 512), (1024, 16), (1024, 1000)])
def test_row_split_many_chunks(gpu, n, batch):
    """N > 256: the residuals are staged in 2 .. 4 chunks of 256 columns, a wave's run of tiles crosses several row blocks."""
    import torch

    mu, sigma, P, logdet, X, lik, ref, tol = problem(n, batch, seed=n)
    ll = lik.logpdf(X)
    assert np.all(np.abs(ll - ref) <= tol), (np.max(np.abs(ll - ref)), tol.min())
    lik.set_form("sweep")
    sw = lik.logpdf(X)
    assert lik.set_form("auto") == "sweep"
    assert np.all(np.abs(sw - ll) <= tol)
    assert batch < 64 or not np.array_equal(sw, ll)          # really another kernel
    # device-resident, padded leading dimension, NaN in the padding
    ld = n + 5
    Xd = torch.full((batch, ld), np.nan, dtype=torch.float64, device=gpu)
    Xd[:, :n] = torch.as_tensor(X, device=gpu)
    out = torch.empty(batch, dtype=torch.float64, device=gpu)
    for _ in range(2):
        out.zero_()
        M._capi.check(M._capi.lib().mcd_mvn_logpdf_batch(lik._h, Xd.data_ptr(), ld, batch, 1, None, out.data_ptr()))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), ll)
    # every number of row groups gives the oracle's value; with the groups of a tile on different XCDs the same bits
    for G in (8, 16, 32):
        with env(MCD_SPLIT_G=G):
            a = lik.logpdf(X)
            with env(MCD_SPLIT_SCATTER=1):
                b = lik.logpdf(X)
        assert np.all(np.abs(a - ref) <= tol) and np.array_equal(a, b), G
    if batch > 80:
        assert np.array_equal(lik.logpdf(X[:70]), lik.logpdf(X)[:70]) or True   # (G may differ with the batch: not bit-equal)
        with env(MCD_SPLIT_G=8):
            assert np.array_equal(lik.logpdf(X[:70]), lik.logpdf(X)[:70])       # same G: a chain's value does not depend on the batch


@pytest.mark.parametrize("leaves,batch", [(66, 40), (98, 512), (128, 512), (129, 96), (129, 512), (200, 33), (257, 100), (400, 512), (512, 512), (513, 7)])
def test_row_split_tree_states(gpu, leaves, batch):
    """Tree states through the row-split form: distances computed while staging (app/Probability.hs:201-207), the root
    slot and the root-branch Jacobian by one thread per chain; against the oracle and against the column sweep."""
    topo = S.random_topology(leaves, seed=leaves)
    n = topo.n_nodes - 2
    mu, sigma = S.random_spd_problem(n, seed=leaves)
    P = np.linalg.inv(sigma)
    logdet = np.linalg.slogdet(sigma)[1]
    st = S.random_states(topo, batch, seed=leaves + 1)
    tl = M.MvnLikelihood(M.Full(mu, P, logdet)).bind_tree(topo)
    ll, lj = tl.loglik(st)
    nb = min(batch, 48)
    ref, refj = O.tree_loglik_full_batch(topo.parent, st.heights[:nb], st.rates[:nb], st.time_height[:nb], st.rate_mean[:nb], mu, P, logdet)
    assert np.max(np.abs(ll[:nb] - ref) / np.abs(ref)) <= 1e-11
    assert np.max(np.abs(lj[:nb] - refj) / np.maximum(1.0, np.abs(refj))) <= 1e-12
    tl.mvn.set_form("sweep")
    ll_s, lj_s = tl.loglik(st)
    tl.mvn.set_form("auto")
    assert np.max(np.abs(ll - ll_s) / np.abs(ll_s)) <= 1e-12
    assert np.array_equal(lj, lj_s)                          # the same arithmetic for the Jacobian
    assert not np.array_equal(ll, ll_s)                      # really the other kernel
    with env(MCD_SPLIT_G=16, MCD_SPLIT_SCATTER=1):
        ll_g, lj_g = tl.loglik(st)
    assert np.max(np.abs(ll_g - ll_s) / np.abs(ll_s)) <= 1e-12 and np.array_equal(lj_g, lj)
    # device-resident states
    import torch

    sd = st.to(gpu)
    ll_d, lj_d = tl.loglik(sd)
    torch.cuda.synchronize()
    assert np.array_equal(ll_d.cpu().numpy(), ll) and np.array_equal(lj_d.cpu().numpy(), lj)


@pytest.mark.parametrize("n,batch", [(200, 40), (256, 512), (257, 9), (300, 64), (384, 512), (513, 100), (769, 17), (1000, 130), (1023, 1), (1024, 1000)])
def test_row_split_gradient(gpu, n, batch):
    """ll and d ll / d x = -Sigma^-1 (x - mu) as two triangular products on the row-split schedule (z = W r, then J y = (J W^T J)(J z)
    over the reversed rows): against the oracle and the sweeps; every number of row groups; in place; padded rows untouched."""
    import torch

    mu, sigma, P, logdet, X, lik, ref, tol = problem(n, batch, seed=n)
    kappa = np.linalg.cond(sigma)
    ll, G = lik.grad(X)
    assert np.array_equal(ll, lik.logpdf(X))                 # the same first product, the same bits
    Gref = O.grad_full_batch(mu, P, X[:32])
    gtol = 64 * n * EPS * kappa * np.abs(Gref).max() * 4
    assert np.max(np.abs(G[:32] - Gref)) <= gtol, (np.max(np.abs(G[:32] - Gref)), gtol)
    lik.set_form("sweep")
    ll_s, G_s = lik.grad(X)
    lik.set_form("auto")
    assert np.max(np.abs(G_s - G)) <= gtol and np.all(np.abs(ll_s - ll) <= tol)
    if n <= 768:
        assert not np.array_equal(G_s, G)                    # really another kernel
    else:
        assert np.array_equal(G_s, G)                        # (round 4: above N = 768 the gradient has no sweep form -- it spilled -- and takes the row split whatever is asked)
    for Gn in (8, 16, 32):
        with env(MCD_SPLIT_G=Gn):
            a = lik.grad(X)
            with env(MCD_SPLIT_SCATTER=1):
                b = lik.grad(X)
        assert np.max(np.abs(a[1] - G_s)) <= gtol and np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0]), Gn
    # device-resident, padded leading dimensions with NaN in the padding; then in place (G = X)
    ld = n + 3
    Xd = torch.full((batch, ld), np.nan, dtype=torch.float64, device=gpu)
    Xd[:, :n] = torch.as_tensor(X, device=gpu)
    Gp = torch.full((batch, n + 5), -7.0, dtype=torch.float64, device=gpu)
    llp = torch.empty(batch, dtype=torch.float64, device=gpu)
    lib = M._capi.lib()
    M._capi.check(lib.mcd_mvn_grad_batch(lik._h, Xd.data_ptr(), ld, batch, 1, None, llp.data_ptr(), Gp.data_ptr(), n + 5))
    torch.cuda.synchronize()
    assert np.array_equal(Gp[:, :n].cpu().numpy(), G) and bool((Gp[:, n:] == -7.0).all()) and np.array_equal(llp.cpu().numpy(), ll)
    M._capi.check(lib.mcd_mvn_grad_batch(lik._h, Xd.data_ptr(), ld, batch, 1, None, llp.data_ptr(), Xd.data_ptr(), ld))
    torch.cuda.synchronize()
    assert np.array_equal(Xd[:, :n].cpu().numpy(), G) and bool(torch.isnan(Xd[:, n:]).all()) and np.array_equal(llp.cpu().numpy(), ll)
    _, G0 = lik.grad(mu[None, :])
    assert np.all(G0 == 0.0)


@pytest.mark.parametrize("leaves,batch", [(98, 512), (129, 96), (130, 512), (200, 33), (257, 100), (400, 512), (512, 512), (513, 7)])
def test_row_split_tree_gradient(gpu, leaves, batch):
    """The gradient wrt a tree state: the two products as above, y parked tile-major, then the chain rule (one workgroup per
    chain): oracle values per chain, the sweeps' values for all; outputs may be their own inputs."""
    import torch

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
    tl.mvn.set_form("sweep")
    sw = tl.grad(st)
    tl.mvn.set_form("auto")
    sc = max(np.abs(sw[1]).max(), np.abs(sw[2]).max())
    assert np.max(np.abs(sw[1] - out[1])) <= 1e-10 * sc and np.max(np.abs(sw[2] - out[2])) <= 1e-10 * sc
    assert np.max(np.abs(sw[3] - out[3]) / np.abs(sw[3])) <= 1e-9 and np.max(np.abs(sw[4] - out[4]) / np.abs(sw[4])) <= 1e-9
    assert np.array_equal(sw[1], out[1]) == (n > 768)        # (above N = 768 the tree gradient has no sweep form since round 4: the row split whatever is asked)
    with env(MCD_SPLIT_G=16, MCD_SPLIT_SCATTER=1):
        og = tl.grad(st)
    assert np.max(np.abs(og[1] - out[1])) <= 1e-10 * sc and np.max(np.abs(og[2] - out[2])) <= 1e-10 * sc
    # device-resident; then the height gradient over the heights and the rate gradient over the rates
    sd = st.to(gpu)
    out_d = tl.grad(sd)
    for a, b in zip(out, out_d):
        assert np.array_equal(a, b.cpu().numpy())
    H, R = sd.heights.clone(), sd.rates.clone()
    llp = torch.empty(batch, dtype=torch.float64, device=gpu)
    gt = torch.empty(batch, dtype=torch.float64, device=gpu)
    gm = torch.empty(batch, dtype=torch.float64, device=gpu)
    M._capi.check(M._capi.lib().mcd_tree_grad_batch(tl._t, H.data_ptr(), R.data_ptr(), H.stride(0), sd.time_height.data_ptr(), sd.rate_mean.data_ptr(),
                                                    batch, 1, None, llp.data_ptr(), H.data_ptr(), R.data_ptr(), gt.data_ptr(), gm.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(H.cpu().numpy(), out[1]) and np.array_equal(R.cpu().numpy(), out[2]) and np.array_equal(gt.cpu().numpy(), out[3])


def test_config5_share_of_one_gpu(gpu):
    """BASELINE config 5 (synthetic 1024-node tree, 4096 chains on 8 GPUs): the share of one GPU, N = 1024 x 512 chains,
    at size -- raw x against the oracle, the size-independent properties, the 1023-dimensional tree (512 leaves) with both
    gradients against the oracle."""
    import torch

    n, batch = 1024, 512
    mu, sigma = S.random_spd_problem(n, seed=1024)
    lik = M.MvnLikelihood.from_covariance(mu, sigma)
    L = lik.cholesky_factor()
    rng = np.random.default_rng(5)
    Z = rng.standard_normal((batch, n))
    X = mu + Z @ L.T
    Xd = torch.as_tensor(X, device=gpu)
    ll = lik.logpdf(Xd).cpu().numpy()
    c = -LN_SQRT_2PI * n
    q = -2.0 * (ll - c) - lik.logdet_sigma
    assert np.max(np.abs(q - np.sum(Z * Z, axis=1)) / np.sum(Z * Z, axis=1)) <= 1e-9       # |L^-1 (x - mu)|^2 = |z|^2
    assert lik.logpdf1(mu) == c - 0.5 * lik.logdet_sigma and np.all(ll <= lik.logpdf1(mu))
    assert np.max(np.abs(lik.logpdf(2 * mu[None, :] - X) - ll) / np.abs(ll)) <= 1e-12       # symmetric around mu
    assert np.array_equal(lik.logpdf(Xd).cpu().numpy(), ll)                                 # repeatable bits
    P = np.linalg.inv(sigma)
    logdet = np.linalg.slogdet(sigma)[1]
    kappa = np.linalg.cond(sigma)
    ref = O.logpdf_full_batch(mu, P, logdet, X[:64])
    assert np.all(np.abs(ll[:64] - ref) <= 64 * n * EPS * kappa * np.maximum(1.0, q[:64]))
    ll_g, G = lik.grad(Xd)
    Gref = O.grad_full_batch(mu, P, X[:16])
    assert np.max(np.abs(G[:16].cpu().numpy() - Gref)) <= 64 * n * EPS * kappa * np.abs(Gref).max() * 4
    assert np.max(np.abs(ll_g.cpu().numpy() - ll) / np.abs(ll)) <= 1e-12
    # the tree of that size: 512 leaves, 1023 nodes, N = 1021
    topo = S.random_topology(512, seed=1024)
    nt = topo.n_nodes - 2
    mu_t, sigma_t = S.random_spd_problem(nt, seed=1023)
    P_t = np.linalg.inv(sigma_t)
    logdet_t = np.linalg.slogdet(sigma_t)[1]
    st = S.random_states(topo, batch, seed=1023)
    tl = M.MvnLikelihood(M.Full(mu_t, P_t, logdet_t)).bind_tree(topo)
    sd = st.to(gpu)
    ll_t, lj_t = tl.loglik(sd)
    out = tl.grad(sd)
    torch.cuda.synchronize()
    ref, refj = O.tree_loglik_full_batch(topo.parent, st.heights[:32], st.rates[:32], st.time_height[:32], st.rate_mean[:32], mu_t, P_t, logdet_t)
    assert np.max(np.abs(ll_t[:32].cpu().numpy() - ref) / np.abs(ref)) <= 1e-11
    assert np.max(np.abs(lj_t[:32].cpu().numpy() - refj) / np.maximum(1.0, np.abs(refj))) <= 1e-12
    assert np.max(np.abs(out[0].cpu().numpy() - ll_t.cpu().numpy()) / np.abs(ll_t.cpu().numpy())) <= 1e-12
    for b in (0, 300, 511):
        gH, gR, gt, gm = O.tree_grad_full(topo.parent, st.heights[b], st.rates[b], st.time_height[b], st.rate_mean[b], mu_t, P_t)
        sc = max(np.abs(gH).max(), np.abs(gR).max())
        assert np.max(np.abs(out[1][b].cpu().numpy() - gH)) <= 1e-9 * sc and np.max(np.abs(out[2][b].cpu().numpy() - gR)) <= 1e-9 * sc
        assert abs(float(out[3][b]) - gt) <= 1e-9 * abs(gt) and abs(float(out[4][b]) - gm) <= 1e-9 * abs(gm)


def _stress(gpu, n, batch, launches, scatter):
    """`launches` back-to-back launches alternating between two input batches (a stale partial sum of the previous launch can
    not pass for the current one), each writing its own output row; every row compared with the first launch of its batch.
    A copy kernel streams 256 MiB beside them on another stream the whole time."""
    import torch

    mu, sigma = S.random_spd_problem(n, seed=n)
    lik = M.MvnLikelihood.from_covariance(mu, sigma)
    XA = torch.as_tensor(S.sample_chains(mu, sigma, batch, seed=1), device=gpu)
    XB = torch.as_tensor(S.sample_chains(mu, sigma, batch, seed=2), device=gpu)
    lib, h = M._capi.lib(), lik._h
    fn = lib.mcd_mvn_logpdf_batch
    side = torch.cuda.Stream()
    src = torch.empty(32 * 1024 * 1024, dtype=torch.float64, device=gpu).normal_()
    dst = torch.empty_like(src)
    block = 2000
    out = torch.empty(block, batch, dtype=torch.float64, device=gpu)
    with env(MCD_SPLIT_SCATTER=int(scatter)):
        expA, expB = lik.logpdf(XA).clone(), lik.logpdf(XB).clone()
        lik.set_form("sweep")
        swA = lik.logpdf(XA)
        lik.set_form("auto")
        assert float(((expA - swA).abs() / swA.abs()).max()) <= 1e-12 and not torch.equal(expA, expB)
        exp = torch.stack([expA, expB]).repeat(block // 2, 1)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        pa, pb, po = XA.data_ptr(), XB.data_ptr(), out.data_ptr()
        done = 0
        while done < launches:
            with torch.cuda.stream(side):
                for _ in range(4):
                    dst.copy_(src, non_blocking=True)
            for i in range(block):
                rc = fn(h, pa if (i & 1) == 0 else pb, n, batch, 1, st, po + i * batch * 8)
                assert rc == 0
            bad = int((out != exp).sum())
            assert bad == 0, f"{bad} wrong values after {done} launches (scatter={scatter})"
            done += block
        torch.cuda.synchronize()


@pytest.mark.parametrize("scatter", [0, 1])
def test_handover_stress_512_chains(gpu, scatter):
    """> 10^6 launches in total over the four stress tests; the headline shape, 256 x 512."""
    _stress(gpu, 256, 512, 300_000, scatter)


@pytest.mark.parametrize("scatter", [0, 1])
def test_handover_stress_one_chain(gpu, scatter):
    """256 x 1: one tile, eight workgroups, the shortest possible gap between the groups' hand-overs."""
    _stress(gpu, 256, 1, 300_000, scatter)


def test_graph_replay_on_another_stream(gpu):
    """A graph captured on stream A owns its scratch: replayed on stream B while eager launches with other inputs go on on A,
    both sequences keep their own results."""
    import torch

    n, batch = 256, 512
    mu, sigma = S.random_spd_problem(n, seed=3)
    lik = M.MvnLikelihood.from_covariance(mu, sigma)
    XA = torch.as_tensor(S.sample_chains(mu, sigma, batch, seed=1), device=gpu)
    XB = torch.as_tensor(S.sample_chains(mu, sigma, batch, seed=2), device=gpu)
    expA, expB = lik.logpdf(XA).clone(), lik.logpdf(XB).clone()
    fn, h = M._capi.lib().mcd_mvn_logpdf_batch, lik._h
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    reps = 50
    outG = torch.zeros(reps, batch, dtype=torch.float64, device=gpu)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=sA):
        st = C.c_void_p(sA.cuda_stream)
        for i in range(reps):
            assert fn(h, XA.data_ptr(), n, batch, 1, st, outG.data_ptr() + i * batch * 8) == 0
    outE = torch.zeros(4000, batch, dtype=torch.float64, device=gpu)
    torch.cuda.synchronize()
    for round_ in range(20):
        outG.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(sB):
            g.replay()
        st = C.c_void_p(sA.cuda_stream)
        for i in range(200):
            assert fn(h, XB.data_ptr(), n, batch, 1, st, outE.data_ptr() + (round_ * 200 + i) * batch * 8) == 0
        torch.cuda.synchronize()
        assert bool((outG == expA).all()), round_
    assert bool((outE == expB).all())


def test_forms_pinned_per_handle(gpu):
    """mcd_mvn_set_form: two handles of one process pin different forms; the process default still steers handles left on auto."""
    n, batch = 256, 64
    mu, sigma = S.random_spd_problem(n, seed=4)
    X = S.sample_chains(mu, sigma, batch, seed=4)
    a, b, c = (M.MvnLikelihood.from_covariance(mu, sigma) for _ in range(3))
    assert a.set_form("sweep") == "auto" and b.set_form("multiply") == "auto"
    la, lb, lc = a.logpdf(X), b.logpdf(X), c.logpdf(X)
    assert not np.array_equal(la, lb) and not np.array_equal(la, lc) and not np.array_equal(lb, lc)   # sweep, k_wide, k_split
    assert np.max(np.abs(la - lb) / np.abs(la)) <= 1e-12 and np.max(np.abs(la - lc) / np.abs(la)) <= 1e-12
    M.set_logpdf_form("sweep")
    try:
        assert np.array_equal(c.logpdf(X), la) and np.array_equal(b.logpdf(X), lb)     # c follows the default, b keeps its own
    finally:
        M.set_logpdf_form("auto")
    assert a.set_form("auto") == "sweep"
    assert np.array_equal(a.logpdf(X), lc)
    with pytest.raises(ValueError):
        a.set_form("fast")
