"""CPU tests of the Metropolis-Hastings twin (oracle/mh_oracle.c) and of the host-side proposal table
(mcmc-date_amd/sampler.py; pure numpy, no device calls).

The reference holds no expected outputs for its proposals (parity unpinned for SURVEY.md 8f row f2); what pins the
twin here: the Random123 known answers of Philox4x32-10, scipy's truncated normal / gamma / erfinv, the closed
forms of every proposal's ratio and Jacobian, and reversibility (the reverse move has the inverse ratio)."""
import dataclasses

import numpy as np
import pytest
import scipy.special as sp
import scipy.stats as st

import mcmc_date_amd as M
import oracle as O
from oracle import prepare as P
from mcmc_date_amd import sampler as SM


def model_for(fx, clock="UncorrelatedGamma", calibrated=None):
    topo = M.Topology(fx["parent"])
    cal = [(int(r[0]), r[2] if r[1] else None, r[3], r[5] if r[4] else None, r[6]) for r in fx["cal"]]
    con = [(int(r[0]), int(r[1]), r[2]) for r in fx["con"]]
    br = [M.Brace(f"b{i}", [int(x) for x in fx["brace_nodes"][fx["brace_ptr"][i]:fx["brace_ptr"][i + 1]]], float(s))
          for i, s in enumerate(fx["brace_sd"])]
    spec = O.PriorSpec(fx["parent"], float(fx["prior_ht"]), clock, cal, con, [(b.nodes, b.sd) for b in br])
    ps, missing = M.proposals(topo, br, calibrations_available=(len(cal) > 0) if calibrated is None else calibrated)
    return topo, ps, missing, O.MhModel(fx["parent"], fx["mu"], fx["sigma_inv"], float(fx["logdet"]), spec, M.table_arrays(ps))


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert O.philox4x32([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox4x32([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox4x32([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    u = np.array([O.uniform_pair(3, c, s, 0) for c in range(50) for s in range(50)]).ravel()
    assert 0 < u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01 and st.kstest(u, "uniform").pvalue > 1e-3
    x = O.philox4x32([5, 7, 11, 0], [3, 0])
    a = ((x[0] << 32 | x[1]) >> 11) + 0.5
    assert O.uniform_pair(3, 7, 11, 5)[0] == a * 2.0 ** -53


def test_special_functions_against_scipy():
    rng = np.random.default_rng(1)
    ys = np.concatenate([rng.uniform(-1, 1, 500), 1 - 10.0 ** -rng.uniform(1, 15.9, 200), -1 + 10.0 ** -rng.uniform(1, 15.9, 200)])
    assert max(abs(O.erfinv(y) - sp.erfinv(y)) / abs(sp.erfinv(y)) for y in ys) < 2e-15
    for m, s, a, b in [(0.3, 0.05, 0.1, 0.9), (0.5, 2.0, 0.0, 1.0), (1.0, 0.01, 0.2, np.inf), (0.0, 0.3, -0.2, 0.05)]:
        d = st.truncnorm((a - m) / s, (b - m) / s, loc=m, scale=s)
        for x in np.linspace(max(a, m - 3 * s), min(b, m + 3 * s), 7):
            assert abs(O.tn_logpdf(m, s, a, b, x) - d.logpdf(x)) < 1e-10
        for p in (0.01, 0.3, 0.5, 0.9, 0.999):
            assert abs(O.tn_quantile(m, s, a, b, p) - d.ppf(p)) < 1e-9 * max(1, abs(d.ppf(p)))
        assert O.tn_quantile(m, s, a, b, 0.0) == a and O.tn_logpdf(m, s, a, b, a - 1e-3) == -np.inf
    assert np.isnan(O.tn_logpdf(2.0, 1.0, 0.0, 1.0, 0.5))          # mean out of bounds: `error` upstream
    assert np.isnan(O.tn_quantile(0.5, 0.0, 0.0, 1.0, 0.5))        # sd <= 0: `error` upstream


@pytest.mark.parametrize("shape,scale", [(100.0, 0.01), (10.0, 0.1), (3.3, 0.3), (0.4, 2.5)])
def test_gamma_sampler_distribution(shape, scale):
    x = np.array([O.gamma_draw(17, c, s, shape, scale) for c in range(40) for s in range(100)])
    assert np.all(x > 0) and st.kstest(x, st.gamma(shape, scale=scale).cdf).pvalue > 1e-3


def test_proposal_table_mirrors_the_reference_cycle(golden):
    fx = golden["12-leaves-variable-rate"]
    topo, ps, missing, _ = model_for(fx, calibrated=True)
    n = topo.n_nodes                                  # 23 nodes, 11 inner, 10 inner non-root
    w = SM.weight_n_branches(n)
    assert w == int(np.floor(np.log(23) / np.log(1.3))) == 11
    inner = [v for v in range(1, n) if not topo.leaves[v]]
    l, r = topo.root_children()
    names = [p.name for p in ps]
    assert names[:5] == ["Time birth rate", "Time death rate", "Rate mean", "Rate variance", "Rates and time tree"] and all(p.weight == w for p in ps[:5])
    assert (ps[4].kind, ps[4].p0, ps[4].n1, ps[4].dim, ps[4].jac_root) == (SM.SCALE_RATES_TREE_CONTRA, 0.1, 10, 12, True)
    slides = [p for p in ps if p.kind == SM.SLIDE_NODE]
    assert sorted(p.node for p in slides) == inner and all(p.weight == 5 and p.dim == 1 and p.p0 == 0.01 for p in slides)
    assert all(p.jac_root == (p.node in (l, r)) for p in ps if p.kind in (SM.SLIDE_NODE, SM.SCALE_SUBTREE_TIME, SM.SCALE_BRANCH_RATE, SM.SCALE_SUBTREE_RATE,
                                                                          SM.SLIDE_NODE_CONTRA, SM.SCALE_SUBTREE_CONTRA))
    csl = [p for p in ps if p.kind == SM.SLIDE_NODE_CONTRA]
    assert sorted(p.node for p in csl) == inner and all(p.dim == 4 and p.p0 == 0.1 and 3 <= p.weight <= 8 for p in csl)
    csc = [p for p in ps if p.kind == SM.SCALE_SUBTREE_CONTRA]
    assert sorted(p.node for p in csc) == inner and all(p.n2 == topo.subtree_size(p.node) and p.dim == p.n1 + p.n2 for p in csc)
    sub = [p for p in ps if p.kind == SM.SCALE_SUBTREE_TIME]
    assert sorted(p.node for p in sub) == inner and all(3 <= p.weight <= 8 and p.n1 == p.dim >= 1 for p in sub)
    branches = [p for p in ps if p.kind == SM.SCALE_BRANCH_RATE]
    assert sorted(p.node for p in branches) == list(range(1, n)) and all(p.weight == 3 and p.p0 == 100.0 for p in branches)
    rsub = [p for p in ps if p.kind == SM.SCALE_SUBTREE_RATE]
    assert all(p.n1 == topo.subtree_size(p.node) for p in rsub) and sorted(p.node for p in rsub) == inner
    pulley = [p for p in ps if p.kind == SM.PULLEY]
    assert len(pulley) == (0 if topo.leaves[l] or topo.leaves[r] else 1) and all(p.weight == 6 and p.jac_root for p in pulley)
    assert [p.name for p in ps[-4:]] == ["Time height", "Time height, rate mean", "[R] Time height, Rate tree", "[R] Trees"]
    assert ps[-4].p0 == 3000.0 and (ps[-3].p0, ps[-3].p1, ps[-3].dim) == (10.0, 0.1, 2) and not ps[-3].jac_root and ps[-2].jac_root
    assert (ps[-1].kind, ps[-1].p0, ps[-1].n1, ps[-1].dim, ps[-1].jac_root, ps[-1].weight) == (SM.SLIDE_ROOT_CONTRA, 10.0, 11, 14, True, w)
    assert missing == []                                 # the whole Metropolis-Hastings cycle of the reference is built
    # without calibrations the time height is not moved (app/Definitions.hs:270-271)
    _, ps0, missing0, _ = model_for(fx, calibrated=False)
    assert len(ps0) == len(ps) - 4 and not any(p.kind in (SM.SCALE_CONTRARILY, SM.SLIDE_ROOT_CONTRA) for p in ps0) and missing0 == []
    # braces add one ultrametric and one contrary proposal each (:165, :221)
    fb = golden["24-leaves-braces"]
    _, psb, _, _ = model_for(fb)
    bps = [p for p in psb if p.kind in (SM.SLIDE_BRACE, SM.SLIDE_BRACE_CONTRA)]
    assert [(p.kind, p.node, p.p0, p.weight, p.jac_root) for p in bps] == [(SM.SLIDE_BRACE, 0, 0.01, 5, False), (SM.SLIDE_BRACE_CONTRA, 0, 0.1, 5, False)]
    assert bps[0].dim == 2 and bps[1].dim == 2 + 2 + 4
    sched = M.cycle_schedule(ps, 3, np.random.default_rng(0))
    assert sched.shape == (3, sum(p.weight for p in ps))
    for row in sched:
        assert np.array_equal(np.bincount(row, minlength=len(ps)), [p.weight for p in ps])
    assert not np.array_equal(sched[0], sched[1])


def test_init_with(golden):
    """initWith: ultrametric, height 1, leaves at 0, rates 1 with stem 0 (app/Definitions.hs:96-123)."""
    for name in ("06-leaves-constant-rate", "25-leaves-bastien"):
        fx = golden[name]
        topo = M.Topology(fx["parent"])
        x = M.init_with(topo, fx["mean_lengths"])
        assert x.is_valid(topo) and x.time_tree[0] == 1.0 and np.all(x.time_tree[topo.leaves] == 0.0)
        assert np.all(x.rate_tree[1:] == 1.0) and x.rate_tree[0] == 0.0
        ref = P.init_state(fx["parent"], fx["mean_lengths"])                  # the oracle's independent restatement
        assert np.allclose(x.time_tree, ref["heights"], rtol=1e-14, atol=1e-15) and np.array_equal(x.rate_tree, ref["rates"])
        ln = fx["mean_lengths"].copy()
        ln[1:][ln[1:] == 0] = ln[1:].mean()                                  # zero branches -> the average branch (:113-117)
        inner = [v for v in range(1, topo.n_nodes) if not topo.leaves[v]]
        t = M.height_tree_to_length_tree(topo, x.time_tree)
        scale = t[inner[0]] / ln[inner[0]]
        assert np.allclose(t[inner], ln[inner] * scale, rtol=1e-12)      # inner branches keep their proportions


def test_every_proposal_kind_closed_forms_and_reversibility(golden):
    fx = golden["24-leaves-braces"]
    topo, ps, _, model = model_for(fx, calibrated=True)
    n = topo.n_nodes
    x0 = M.init_with(topo, fx["mean_lengths"])
    rng = np.random.default_rng(3)
    H = x0.time_tree.copy()
    R = np.concatenate([[0.0], rng.uniform(0.6, 1.6, n - 1)])
    sc = np.array([1.3, 0.7, 2.5, 0.9, 0.4])
    size = np.array([topo.subtree_size(v) for v in range(n)])
    seen = set()
    for p_id, p in enumerate(ps):
        t = float(rng.uniform(0.5, 2.0))
        sc1, H1, R1, lnq, lnj = O.propose_once(model, p_id, t, 99, 4, 1000 + p_id, sc, H, R)
        seen.add(p.kind)
        v = p.node
        sub = np.arange(v, v + size[v])
        if p.kind in (SM.SCALE_SCALAR, SM.SCALE_BRANCH_RATE, SM.SCALE_SUBTREE_RATE, SM.SCALE_NORM_TREE, SM.SCALE_VAR_TREE,
                      SM.SCALE_VAR_TREE_AUTO, SM.SCALE_CONTRARILY):
            k, th = p.p0 / t, (p.p1 * t if p.kind == SM.SCALE_CONTRARILY else t / p.p0)
            if p.kind == SM.SCALE_SCALAR:
                u = sc1[v] / sc[v]
                assert np.array_equal(np.delete(sc1, v), np.delete(sc, v)) and np.array_equal(R1, R) and abs(lnj + np.log(u)) < 1e-12
            elif p.kind == SM.SCALE_BRANCH_RATE:
                u = R1[v] / R[v]
                assert np.array_equal(np.delete(R1, v), np.delete(R, v)) and abs(lnj + np.log(u)) < 1e-12
            elif p.kind == SM.SCALE_SUBTREE_RATE:
                u = R1[v] / R[v]
                assert np.allclose(R1[sub], R[sub] * u, rtol=1e-15) and np.array_equal(np.delete(R1, sub), np.delete(R, sub))
                assert abs(lnj - (size[v] - 2) * np.log(u)) < 1e-11
            elif p.kind == SM.SCALE_NORM_TREE:
                u = sc[v] / sc1[v]
                assert np.allclose(R1[1:], R[1:] * u, rtol=1e-14) and R1[0] == R[0] and abs(lnj - (n - 1 - 3) * np.log(u)) < 1e-10
            elif p.kind == SM.SCALE_VAR_TREE:
                u = np.sqrt(sc1[4] / sc[4])
                mu = R[1:].mean()
                assert np.allclose(R1[1:], (R[1:] - mu) * u + mu, rtol=1e-13) and abs(R1[1:].mean() - mu) < 1e-13
                assert abs(lnj - (n - 1) * np.log(u - u / (n - 1) + 1 / (n - 1))) < 1e-10
            elif p.kind == SM.SCALE_VAR_TREE_AUTO:
                u = np.sqrt(sc1[4] / sc[4])
                assert np.allclose(R1[1:], sc[3] + u * (R[1:] - sc[3]), rtol=1e-13) and abs(lnj - (n - 1) * np.log(u)) < 1e-10
            else:
                u = sc1[2] / sc[2]
                assert abs(sc1[3] * u - sc[3]) < 1e-15 and abs(lnj + 2 * np.log(u)) < 1e-12
            g = st.gamma(k, scale=th)
            assert abs(lnq - (g.logpdf(1 / u) - g.logpdf(u))) < 1e-9 * max(1.0, abs(lnq))
            assert np.array_equal(H1, H)
        elif p.kind in (SM.SLIDE_NODE_CONTRA, SM.SCALE_SUBTREE_CONTRA, SM.SLIDE_ROOT_CONTRA, SM.SCALE_RATES_TREE_CONTRA, SM.SLIDE_BRACE,
                        SM.SLIDE_BRACE_CONTRA):
            s1 = t * p.p0
            z = lambda m, a, b: st.norm.cdf((b - m) / s1) - st.norm.cdf((a - m) / s1)
            par = topo.parent
            tl = lambda HH: HH[np.maximum(par, 0)] - HH                       # branch lengths
            if p.kind == SM.SLIDE_NODE_CONTRA:
                ch = topo.children(v)
                assert max(H[c] for c in ch) <= H1[v] <= H[par[v]] and np.array_equal(np.delete(H1, v), np.delete(H, v))
                touched = ch + [v]
                # contrary: every touched branch keeps its time * rate product
                assert np.allclose((tl(H1) * R1)[touched], (tl(H) * R)[touched], rtol=1e-13)
                assert np.array_equal(np.delete(R1, touched), np.delete(R, touched))
                assert abs(lnj - np.sum(np.log(R1[touched] / R[touched]))) < 1e-11
                assert abs(lnq - (np.log(z(H[v], max(H[c] for c in ch), H[par[v]])) - np.log(z(H1[v], max(H[c] for c in ch), H[par[v]])))) < 1e-9
            elif p.kind == SM.SCALE_SUBTREE_CONTRA:
                xi = H1[v] / H[v]
                assert np.allclose(H1[sub[1:]], H[sub[1:]] * xi, rtol=1e-15) and np.array_equal(np.delete(H1, sub), np.delete(H, sub))
                assert np.allclose((tl(H1) * R1)[sub], (tl(H) * R)[sub], rtol=1e-13) and np.array_equal(np.delete(R1, sub), np.delete(R, sub))
                assert abs(lnj - ((p.n1 - p.n2) * np.log(xi) + np.log(R1[v] / R[v]))) < 1e-11
            elif p.kind == SM.SLIDE_ROOT_CONTRA:
                u = sc1[2] / sc[2]
                l, r = topo.root_children()
                assert sc1[2] >= sc[2] * max(H[l], H[r]) and H1[0] == 1.0 and np.allclose(H1[1:], H[1:] / u, rtol=1e-15)
                # absolute node ages below the root and the root branches' time * rate products are unchanged
                assert np.allclose(sc1[2] * H1[1:], sc[2] * H[1:], rtol=1e-14)
                assert np.allclose((sc1[2] * tl(H1) * R1)[[l, r]], (sc[2] * tl(H) * R)[[l, r]], rtol=1e-13)
                assert abs(lnj - (-p.n1 * np.log(u) + np.log(R1[l] / R[l]) + np.log(R1[r] / R[r]))) < 1e-11
                assert np.array_equal(np.delete(R1, [l, r]), np.delete(R, [l, r])) and np.array_equal(np.delete(sc1, 2), np.delete(sc, 2))
            elif p.kind == SM.SCALE_RATES_TREE_CONTRA:
                l, r = topo.root_children()
                xi = max(H1[l], H1[r]) / max(H[l], H[r])
                assert H1[0] == H[0] and np.allclose(H1[1:], H[1:] * xi, rtol=1e-15) and np.array_equal(R1, R)
                assert abs(sc1[0] * xi - sc[0]) < 1e-15 and abs(sc1[3] * xi - sc[3]) < 1e-15 and abs(lnj - (p.n1 - 3) * np.log(xi)) < 1e-11
            else:
                nodes = [36, 6]
                delta = H1[nodes[0]] - H[nodes[0]]
                assert delta != 0 and np.allclose(H1[nodes] - H[nodes], delta, rtol=1e-12) and np.array_equal(np.delete(H1, nodes), np.delete(H, nodes))
                if p.kind == SM.SLIDE_BRACE:
                    assert np.array_equal(R1, R) and lnj == 0.0
                else:
                    touched = nodes + [c for x in nodes for c in topo.children(x)]
                    assert np.allclose((tl(H1) * R1)[touched], (tl(H) * R)[touched], rtol=1e-13)
                    assert abs(lnj - np.sum(np.log(R1[touched] / R[touched]))) < 1e-11
            assert M.State(1, 1, 1, H1, 1, 1, np.maximum(R1, 1e-9)).is_valid(topo)
        else:
            s1 = t * p.p0
            z = lambda m, a, b: st.norm.cdf((b - m) / s1) - st.norm.cdf((a - m) / s1)
            if p.kind == SM.SLIDE_NODE:
                a, b = max(H[c] for c in topo.children(v)), H[topo.parent[v]]
                assert a <= H1[v] <= b and np.array_equal(np.delete(H1, v), np.delete(H, v)) and lnj == 0.0
                assert abs(lnq - (np.log(z(H[v], a, b)) - np.log(z(H1[v], a, b)))) < 1e-9
            elif p.kind == SM.SCALE_SUBTREE_TIME:
                xi = H1[v] / H[v]
                assert 0 < H1[v] <= H[topo.parent[v]] and np.allclose(H1[sub[1:]], H[sub[1:]] * xi, rtol=1e-15)
                assert np.array_equal(np.delete(H1, sub), np.delete(H, sub)) and abs(lnj - (p.n1 - 1) * np.log(xi)) < 1e-11
                assert abs(lnq - (np.log(z(H[v], 0.0, H[topo.parent[v]])) - np.log(z(H1[v], 0.0, H[topo.parent[v]])))) < 1e-9
            else:
                l, r = topo.root_children()
                u = H[l] - H1[l]
                assert abs((H1[r] - H[r]) - u) < 1e-15 and H1[0] == H[0]
                assert abs(lnj - ((p.n1 - 1) * np.log(H1[l] / H[l]) + (p.n2 - 1) * np.log(H1[r] / H[r]))) < 1e-11
            assert np.array_equal(R1, R) and np.array_equal(sc1, sc)
            assert M.State(1, 1, 1, H1, 1, 1, np.maximum(R1, 1e-9)).is_valid(topo)     # still a valid ultrametric tree
    assert seen == set(range(16))


def test_chain_is_reproducible_and_chain_offset_selects_the_stream(golden):
    fx = golden["12-leaves-variable-rate"]
    topo, ps, _, model = model_for(fx)
    x0 = M.init_with(topo, fx["mean_lengths"])
    x0.time_height = float(fx["prior_ht"])
    s = M.StateBatch.from_states([x0] * 6)
    mk = lambda lo, hi, chain0: O.MhChains(model, s.time_birth_rate[lo:hi], s.time_death_rate[lo:hi], s.time_height[lo:hi],
                                           s.heights[lo:hi], s.rate_mean[lo:hi], s.rate_variance[lo:hi], s.rates[lo:hi], seed=5, chain0=chain0)
    sched = M.cycle_schedule(ps, 5, np.random.default_rng(0))
    full, again, shard = mk(0, 6, 0), mk(0, 6, 0), mk(4, 6, 4)
    full.run(sched); again.run(sched); shard.run(sched)
    assert np.array_equal(full.H, again.H) and np.array_equal(full.acc, again.acc)
    assert np.array_equal(full.H[4:], shard.H) and np.array_equal(full.R[4:], shard.R) and np.array_equal(full.acc[4:], shard.acc)
    assert not np.array_equal(full.H[0], full.H[1])                       # chains differ from each other
    # two stretches equal one long stretch (the step counter continues)
    two = mk(0, 6, 0)
    two.run(sched[:2]); two.run(sched[2:])
    assert np.array_equal(two.H, full.H) and np.array_equal(two.tried, full.tried)


def test_auto_tuning_rule(golden):
    fx = golden["06-leaves-constant-rate"]
    topo, ps, _, model = model_for(fx)
    x0 = M.init_with(topo, fx["mean_lengths"])
    s = M.StateBatch.from_states([x0] * 2)
    ch = O.MhChains(model, s.time_birth_rate, s.time_death_rate, s.time_height, s.heights, s.rate_mean, s.rate_variance, s.rates, seed=1)
    P = len(ps)
    ch.tried[:] = 100
    ch.acc[0] = 44
    ch.acc[1] = 100
    ch.tune[1, 0] = 900.0
    dims = np.array([p.dim for p in ps])
    opt = np.select([dims == 1, dims == 2, dims == 3, dims == 4, dims == 5], [0.44, 0.352, 0.316, 0.279, 0.275], 0.234)
    ch.autotune()
    assert np.allclose(ch.tune[0], np.exp(2 * (0.44 - opt)), rtol=1e-14)
    assert ch.tune[1, 0] == 1e3 and np.allclose(ch.tune[1, 1:], np.exp(2 * (1.0 - opt[1:])), rtol=1e-14)      # clamped at 1e3
    assert not ch.tried.any() and not ch.acc.any()
    ch.autotune()                                                              # nothing tried: parameters stay
    assert ch.tune[1, 0] == 1e3


def test_twin_samples_a_known_target():
    """A target with known marginals: a 3-leaf tree with a flat likelihood, so the chain samples the prior.  The
    uncorrelated gamma clock prior is a normalised density of the rates for every rVar, hence the marginal of rVar is
    its hyper-prior gamma(3/2, 1/6) (mean 1/4) and that of rMu is exponential(ht = 1) (mean 1).  A wrong proposal
    ratio or Jacobian in any move that touches rVar, rMu or the rates shifts these means."""
    parent = np.array([-1, 0, 1, 1, 0], np.int32)
    n = 3
    mu = np.array([1.0, 0.4, 0.4])
    sigma_inv = np.eye(n) * 1e-12                                  # flat likelihood
    spec = O.PriorSpec(parent, 1.0, "UncorrelatedGamma", [], [], [])
    topo = M.Topology(parent)
    ps, _ = M.proposals(topo, [], False)
    # the same moves without the root-branch Jacobian lift: the target then is exactly the prior
    ps = [dataclasses.replace(p, jac_root=False) for p in ps]
    model = O.MhModel(parent, mu, sigma_inv, 0.0, spec, M.table_arrays(ps))
    B = 16
    x0 = M.State(1.0, 1.0, 1.0, np.array([1.0, 0.4, 0.0, 0.0, 0.0]), 1.0, 1.0, np.array([0.0, 1.0, 1.0, 1.0, 1.0]))
    s = M.StateBatch.from_states([x0] * B)
    ch = O.MhChains(model, s.time_birth_rate, s.time_death_rate, s.time_height, s.heights, s.rate_mean, s.rate_variance, s.rates, seed=77)
    rng = np.random.default_rng(0)
    for period in (50, 50, 100, 100, 200):
        ch.run(M.cycle_schedule(ps, period, rng))
        ch.autotune()
    draws = []
    for _ in range(400):
        ch.run(M.cycle_schedule(ps, 5, rng))
        draws.append((ch.rVar.copy(), ch.rMu.copy()))
    rvar = np.array([d[0] for d in draws]).ravel()
    rmu = np.array([d[1] for d in draws]).ravel()
    # hyper-priors: rVar ~ gamma(3/2, 1/6) (mean 0.25), rMu ~ exponential(ht = 1) (mean 1); rates have a proper density
    assert abs(rvar.mean() - 0.25) < 0.03, rvar.mean()
    assert abs(rmu.mean() - 1.0) < 0.12, rmu.mean()
    rate = ch.acc.sum(0) / np.maximum(1, ch.tried.sum(0))
    assert np.all((rate > 0.05) & (rate < 0.95))


def _prior_chain_moments(ps, n_rounds=500, seed=5):
    """Flat likelihood, soft calibration of the root, no root-branch lift: the chain samples the prior."""
    parent = np.array([-1, 0, 1, 1, 0], np.int32)
    spec = O.PriorSpec(parent, 1.0, "UncorrelatedGamma", [(0, 0.5, 0.025, 2.0, 0.025)], [], [])
    ps = [dataclasses.replace(p, jac_root=False) for p in ps]
    model = O.MhModel(parent, np.array([1.0, 0.4, 0.4]), np.eye(3) * 1e-12, 0.0, spec, M.table_arrays(ps))
    B = 64
    x0 = M.State(1.0, 1.0, 1.0, np.array([1.0, 0.4, 0.0, 0.0, 0.0]), 1.0, 1.0, np.array([0.0, 1.0, 1.0, 1.0, 1.0]))
    s = M.StateBatch.from_states([x0] * B)
    ch = O.MhChains(model, s.time_birth_rate, s.time_death_rate, s.time_height, s.heights, s.rate_mean, s.rate_variance, s.rates, seed=seed)
    rng = np.random.default_rng(0)
    for period in (50, 50, 100, 100, 200, 200):
        ch.run(M.cycle_schedule(ps, period, rng))
        ch.autotune()
    rv, rr = [], []
    for _ in range(n_rounds):
        ch.run(M.cycle_schedule(ps, 10, rng))
        rv.append(ch.rVar.copy())
        rr.append(ch.R[:, [1, 4]].copy())
    rv, rr = np.array(rv), np.array(rr)
    return rv.mean(), rv.mean(axis=0).std() / np.sqrt(B), rr.mean(), rr.mean(axis=(0, 2)).std() / np.sqrt(B)


def test_two_reference_jacobians_are_not_determinants():
    """Finding about the reference, kept visible: scaleVarianceAndTree (Unconstrained.hs:321-326) uses the product of
    the diagonal of its Jacobian matrix, (u - u/n + 1/n)^n, where the determinant of the mean-preserving map is u^(n-1);
    slideRootContrarily (Contrary.hs:160-170) uses u^-n with n counting the root, where n - 1 heights are divided by u.
    With a flat likelihood the chain must reproduce the prior: E[rVar] = 1/4 and E[rate] = 1.  The restated reference
    cycle misses both by many standard errors, the cycle with the determinants (proposals(..., exact_jacobians=True))
    does not.  The default stays the reference's behaviour (parity)."""
    topo = M.Topology(np.array([-1, 0, 1, 1, 0], np.int32))
    ref_cycle, _ = M.proposals(topo, [], True)
    exact_cycle, _ = M.proposals(topo, [], True, exact_jacobians=True)
    assert [p.kind for p in ref_cycle] == [p.kind for p in exact_cycle]
    changed = [(a.kind, a.n1, a.p1, b.n1, b.p1) for a, b in zip(ref_cycle, exact_cycle) if a != b]
    assert changed == [(SM.SCALE_VAR_TREE, 0, 0.0, 0, 1.0), (SM.SLIDE_ROOT_CONTRA, 2, 0.0, 1, 0.0)]
    m_ref, se_ref, r_ref, ser_ref = _prior_chain_moments(ref_cycle)
    m_ex, se_ex, r_ex, ser_ex = _prior_chain_moments(exact_cycle)
    assert abs(m_ex - 0.25) < 4 * se_ex + 0.002 and abs(r_ex - 1.0) < 4 * ser_ex + 0.004, (m_ex, se_ex, r_ex, ser_ex)
    assert m_ref < 0.25 - 5 * se_ref and r_ref > 1.0 + 5 * ser_ref, (m_ref, se_ref, r_ref, ser_ref)
