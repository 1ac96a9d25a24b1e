"""`prepare` (row f4, data contract): the product's host-side restatement against the oracle's independent
one, on the reference's own tree lists.  Needs /root/reference (build container only): skipped elsewhere."""
import os

import numpy as np
import pytest

import mcmc_date_amd as M
from mcmc_date_amd import prepare as PP

REF = "/root/reference/tests"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference inputs are not available on this machine")

CASES = {
    "06-leaves-constant-rate": ("data/test.treelist", "data/time.tree"),
    "12-leaves-variable-rate": ("data/test.treelist", "data/time.alpha.rotated.tree"),
    "24-leaves-braces": ("data/test.treelist", "data/time.relabelled.tree"),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_prepare_matches_fixture_operands(golden, name, tmp_path):
    tl, rt = CASES[name]
    p = PP.prepare(os.path.join(REF, name, tl), os.path.join(REF, name, rt))
    fx = golden[name]
    assert p.n_trees - p.n_burn_in == int(fx["n_trees_kept"])
    assert np.array_equal(p.topology.parent, fx["parent"])
    assert np.allclose(p.mu, fx["mu"], rtol=1e-14, atol=0) and np.allclose(p.sigma, fx["sigma"], rtol=1e-12, atol=1e-18)
    assert isinstance(p.lhd, M.Full)
    assert np.allclose(p.lhd.sigma_inv, fx["sigma_inv"], rtol=1e-9) and abs(p.lhd.logdet_sigma - float(fx["logdet"])) < 1e-9
    assert np.allclose(p.mean_lengths, fx["mean_lengths"], rtol=1e-14, atol=0)
    # files: .data round trip and a parseable mean tree with index labels on inner nodes
    PP.write_prepared(str(tmp_path / "t"), p)
    back = M.read_data_file(str(tmp_path / "t.data"))
    assert isinstance(back, M.Full) and np.array_equal(back.mu, p.mu)
    topo2, ln2 = M.parse_newick(open(tmp_path / "t.meantree").read())
    assert np.array_equal(topo2.parent, p.topology.parent) and np.allclose(ln2, p.mean_lengths, rtol=1e-15)
    assert topo2.names[0] == "0" and all(n for n in topo2.names)


def test_prepare_bastien_annotated_newick():
    """25-leaves-bastien: BEAST-style annotated Newick ([&index=..] comments), 3001 trees."""
    import oracle.prepare as OP

    d = os.path.join(REF, "25-leaves-bastien", "data")
    p = PP.prepare(os.path.join(d, "alignment.fasta.trees.only"), os.path.join(d, "time.tree"))
    assert len(p.mu) == 47 and p.n_trees == 3001 and p.n_burn_in == 500
    o = OP.prepare(os.path.join(d, "alignment.fasta.trees.only"), os.path.join(d, "time.tree"))
    assert np.allclose(p.mu, o.mu, rtol=1e-14) and np.allclose(p.sigma, o.sigma, rtol=1e-12, atol=1e-18)
    assert abs(p.lhd.logdet_sigma - o.logdet) < 1e-9 and np.all(np.linalg.eigvalsh(p.sigma) > 0)


def test_prepare_reroots_when_needed(tmp_path):
    """A rooted tree with another root: every tree of the list is re-rooted at its outgroup."""
    tl, rt = CASES["06-leaves-constant-rate"]
    a = os.path.join(REF, "06-leaves-constant-rate", tl)
    p0 = PP.prepare(a, os.path.join(REF, "06-leaves-constant-rate", rt))
    alt = tmp_path / "alt.tree"
    alt.write_text("((a:1,b:1):1,(c:1,(d:1,(e:1,f:1):1):1):1);")
    p1 = PP.prepare(a, str(alt))
    rooted, _ = M.read_newick_file(str(alt))[0]
    assert PP.root_bipartition(p1.topology) == PP.root_bipartition(rooted)
    assert PP.root_bipartition(p1.topology) != PP.root_bipartition(p0.topology)
    assert len(p1.mu) == len(p0.mu) == 9
    # the total tree length is invariant under re-rooting, tree by tree and hence in the mean
    assert abs(p1.mu.sum() - p0.mu.sum()) < 1e-12 and abs(p1.mean_lengths.sum() - p0.mean_lengths.sum()) < 1e-12
    # the outgroup side comes first, and the root branch is split in half
    l, r = p1.topology.root_children()
    assert sorted(n for n in p1.topology.names[l:r] if n) == ["a", "b"] and p1.mean_lengths[l] == p1.mean_lengths[r]
    # an outgroup that is not a clade of the listed trees is a structural fault
    bad = tmp_path / "bad.tree"
    bad.write_text("((a:1,c:1):1,(b:1,(d:1,(e:1,f:1):1):1):1);")
    with pytest.raises(M.TreeError, match="not a clade"):
        PP.prepare(a, str(bad))


def test_prepare_other_specs(tmp_path):
    tl, rt = CASES["06-leaves-constant-rate"]
    a, b = os.path.join(REF, "06-leaves-constant-rate", tl), os.path.join(REF, "06-leaves-constant-rate", rt)
    u = PP.prepare(a, b, "UnivariateNormal")
    assert isinstance(u.lhd, M.Univariate) and np.allclose(u.lhd.vs, np.diag(u.sigma))
    assert isinstance(PP.prepare(a, b, "NoLikelihood").lhd, M.NoData)
    sp = PP.prepare(a, b, "SparseMultivariateNormal 0.1")
    assert isinstance(sp.lhd, M.Sparse) and np.array_equal(sp.lhd.mu, u.mu)
    with pytest.raises(NotImplementedError):
        PP.prepare(a, b, "SomethingElse")


def test_node_prior_loaders_match_fixtures(golden):
    """calibrations.csv / constraints.csv / braces.json of the reference's test directories -> pre-order node ids."""
    for name in ("12-leaves-variable-rate", "24-leaves-braces"):
        fx = golden[name]
        topo = M.Topology(fx["parent"], list(fx["names"]))
        d = os.path.join(REF, name, "data")
        cals = M.load_calibrations(topo, os.path.join(d, "calibrations.csv"))
        assert [c.node for c in cals] == [int(r[0]) for r in fx["cal"]]
        assert [c.lower for c in cals] == [r[2] for r in fx["cal"]] and [c.upper for c in cals] == [r[5] for r in fx["cal"]]
        cons = M.load_constraints(topo, os.path.join(d, "constraints.csv"))
        assert [(k.young, k.old, k.p) for k in cons] == [(int(r[0]), int(r[1]), r[2]) for r in fx["con"]]
        assert M.get_mean_root_height(cals) == float(fx["prior_ht"])
    br = M.load_braces(topo, os.path.join(d, "braces.json"))
    assert [b.nodes for b in br] == [[int(n) for n in fx["brace_nodes"]]] and br[0].sd == 1e-4
    assert cals[0].name == "CladeRoot" and cals[0].node == 0
    assert M.get_mean_root_height(cals[1:]) is None


def test_graphical_lasso_optimality_conditions():
    """`prepare` with SparseMultivariateNormal (app/Main.hs:257-276) needs the graphical lasso, third-party Fortran in the
    reference; ours is checked against the conditions that characterise the unique optimum (Friedman et al. 2008, eq. 2.4):
    W Theta = I; W_ij - S_ij = rho sign(Theta_ij) where Theta_ij != 0, |W_ij - S_ij| <= rho where it is 0; W_ii = S_ii (+ rho);
    and against scikit-learn's solver where that is installed."""
    from mcmc_date_amd.prepare import graphical_lasso

    rng = np.random.default_rng(0)
    X = rng.standard_normal((9, 11))                       # fewer samples than dimensions, as in the mtCDNApri analysis
    X[:, 3] += X[:, 2]
    X[:, 7] -= 0.7 * X[:, 1]
    S = np.corrcoef(X, rowvar=False)
    for rho in (0.05, 0.1, 0.3):
        for pen in (True, False):
            W, T = graphical_lasso(S, rho, penalize_diagonal=pen)
            assert np.abs(W @ T - np.eye(11)).max() <= 1e-8
            assert np.abs(np.diag(W) - np.diag(S) - (rho if pen else 0.0)).max() <= 1e-12
            off = ~np.eye(11, dtype=bool)
            nz = (T != 0) & off
            assert np.abs((W - S)[nz] - rho * np.sign(T[nz])).max() <= 1e-7
            assert np.all(np.abs((W - S)[off & ~nz]) <= rho + 1e-9)
            assert np.linalg.eigvalsh(T).min() > 0
        try:
            from sklearn.covariance import graphical_lasso as sk
        except Exception:
            continue
        cov, prec = sk(S, alpha=rho, tol=1e-10, max_iter=5000)
        W, T = graphical_lasso(S, rho, penalize_diagonal=False)
        assert np.abs(W - cov).max() <= 1e-8 and np.abs(T - prec).max() <= 1e-7
    assert (graphical_lasso(S, 0.3)[1] == 0).sum() > (graphical_lasso(S, 0.05)[1] == 0).sum()     # more penalty, sparser


def test_prepare_sparse_on_the_references_mtcdnapri_trees(tmp_path):
    """`./run ... s p` on the ten PhyloBayes trees of the reference's mtCDNApri analysis (nine after the burn-in: fewer than
    the eleven branches, the sample covariance is singular and only the penalised estimate exists)."""
    import json
    import os

    from mcmc_date_amd.likelihood import Sparse
    from mcmc_date_amd.prepare import prepare

    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mtCDNApri_prior_samples.json")))
    paths = {}
    for k in ("rooted_tree", "tree_list"):
        paths[k] = str(tmp_path / k)
        open(paths[k], "w").write(fx["inputs"][k])
    p = prepare(paths["tree_list"], paths["rooted_tree"], "SparseMultivariateNormal 0.1")
    assert isinstance(p.lhd, Sparse) and len(p.mu) == 11
    n = 11
    P = np.zeros((n, n))
    for (i, j), v in p.lhd.sigma_inv_assoc:
        P[i, j] = v
    assert np.allclose(P, P.T) and np.linalg.eigvalsh(P).min() > 0 and 40 < len(p.lhd.sigma_inv_assoc) < 121
    assert abs(np.linalg.slogdet(np.linalg.inv(P))[1] - p.lhd.logdet_sigma) <= 1e-6 * abs(p.lhd.logdet_sigma)
    assert np.linalg.matrix_rank(p.sigma, tol=1e-12) < n                     # the sample covariance itself is singular
