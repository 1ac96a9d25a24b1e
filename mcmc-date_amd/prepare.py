"""`prepare`: tree list -> operands of the likelihood closure (one-time host step).

Mirrors the reference's `prepare` mode (app/Main.hs:159-307): read a posterior sample of trees with
branch lengths, drop the first `nTrees div 6` as burn-in, root every tree at the outgroup of the given
rooted tree, check that all topologies (including sub-tree order) agree, build the row matrix of branch
vectors `sumFirstTwo . getBranches` (app/Main.hs:101-104), take mean and (n-1)-normalised covariance
(`meanCov`), invert (`invlndet`), and write `<name>.data` (aeson encoding of `LikelihoodDataStore`,
app/Main.hs:75-81, 240, 286) and `<name>.meantree` (Newick, node labels replaced by indices when they
are not alphabetic, app/Main.hs:288-307, app/Tools.hs:75-81).

This is host code by design (it runs once per analysis; the reference does it on the CPU through
hmatrix/LAPACK); the per-step path is the HIP kernels.  Row SURVEY.md 8(f) f4 ("data contract").

Re-rooting: the reference calls elynx-tree's `outgroup` [third party].  When a tree of the list is already
rooted at the rooted tree's bipartition (every tree list shipped in the reference's tests/ except
25-leaves-bastien) the tree is kept as it is.  Otherwise it is re-rooted on the branch that separates the
outgroup: the new root's first child is the outgroup side, the branch is split in half (the split point is
erased by `sumFirstTwo` for mu/Sigma and only positions the root of the mean tree), and the remaining
sub-tree orders are those obtained by walking away from the new root (old parent last).
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from .likelihood import Full, LikelihoodData, NoData, Sparse, Univariate, write_data_file
from .tree import Topology, TreeError, get_branches, read_newick_file, sum_first_two


@dataclass
class Prepared:
    lhd: LikelihoodData
    mu: np.ndarray
    sigma: np.ndarray
    topology: Topology
    mean_lengths: np.ndarray          # mean branch length per pre-order node (both root branches kept apart)
    n_trees: int
    n_burn_in: int

    def mean_tree_newick(self) -> str:
        return to_newick(self.topology, self.mean_lengths, index_labels=True)


# ---------------------------------------------------------------------------------------------
# small tree utilities on (Topology, lengths)
# ---------------------------------------------------------------------------------------------
def _leaf_sets(topo: Topology) -> List[frozenset]:
    sets: List[Optional[frozenset]] = [None] * topo.n_nodes
    for v in range(topo.n_nodes - 1, -1, -1):
        ch = topo.children(v)
        sets[v] = frozenset([topo.names[v]]) if not ch else frozenset().union(*[sets[c] for c in ch])
    return sets  # type: ignore


def root_bipartition(topo: Topology) -> frozenset:
    l, r = topo.root_children()
    s = _leaf_sets(topo)
    return frozenset([s[l], s[r]])


def to_newick(topo: Topology, lengths: np.ndarray, index_labels: bool = False) -> str:
    def label(v):
        nm = topo.names[v]
        if index_labels and (nm == "" or re.fullmatch(r"\d+", nm)):   # assignIndices, app/Tools.hs:75-81
            return str(v)
        return nm

    def rec(v):
        ch = topo.children(v)
        s = "(" + ",".join(rec(c) for c in ch) + ")" if ch else ""
        return f"{s}{label(v)}:{float(lengths[v])!r}"

    return rec(0) + ";"


def reroot_at_outgroup(topo: Topology, lengths: np.ndarray, outgroup: frozenset) -> Tuple[Topology, np.ndarray]:
    """Root the tree on the branch that separates `outgroup` (a set of leaf names) from the rest."""
    s = _leaf_sets(topo)
    all_leaves = s[0]
    if root_bipartition(topo) == frozenset([outgroup, all_leaves - outgroup]):
        return topo, lengths
    # unrooted adjacency with branch lengths (a bifurcating root is dissolved)
    nn = topo.n_nodes
    adj = {v: [] for v in range(nn)}
    for v in range(1, nn):
        adj[v].append((int(topo.parent[v]), float(lengths[v])))
        adj[int(topo.parent[v])].append((v, float(lengths[v])))
    rc = topo.children(0)
    if len(rc) == 2:
        a, b = rc
        w = float(lengths[a] + lengths[b])
        adj[a] = [(x, l) for x, l in adj[a] if x != 0] + [(b, w)]
        adj[b] = [(x, l) for x, l in adj[b] if x != 0] + [(a, w)]
        del adj[0]
    # the edge whose far side is exactly the outgroup
    target = None
    for v in range(1, nn):
        if v in adj and s[v] in (outgroup, all_leaves - outgroup) and int(topo.parent[v]) in adj:
            target = (v, int(topo.parent[v])) if s[v] == outgroup else (int(topo.parent[v]), v)
            break
    if target is None and len(rc) == 2:
        a, b = rc
        if s[a] in (outgroup, all_leaves - outgroup):
            target = (a, b) if s[a] == outgroup else (b, a)
    if target is None:
        raise TreeError("outgroup: the outgroup is not a clade of the (unrooted) tree")
    og_node, in_node = target
    w = next(l for x, l in adj[og_node] if x == in_node)
    parent, length, names = [-1], [0.0], [""]

    def walk(v, frm, par, ln):
        me = len(parent)
        parent.append(par)
        length.append(ln)
        names.append(topo.names[v])
        for x, l in adj[v]:
            if x != frm:
                walk(x, v, me, l)

    walk(og_node, in_node, 0, w / 2)
    walk(in_node, og_node, 0, w / 2)
    return Topology(np.asarray(parent, np.int32), names), np.asarray(length)


# ---------------------------------------------------------------------------------------------
# prepare
# ---------------------------------------------------------------------------------------------
def prepare(tree_list_path: str, rooted_tree_path: str, likelihood_spec: str = "FullMultivariateNormal") -> Prepared:
    trees_all = read_newick_file(tree_list_path)                          # app/Main.hs:162
    n_trees = len(trees_all)
    n_burn = n_trees // 6                                                 # :166
    trs = trees_all[n_burn:]
    for topo, _ in trees_all:                                             # :170-173
        lv = [n for n, is_leaf in zip(topo.names, topo.leaves) if is_leaf]
        if len(set(lv)) != len(lv):
            raise TreeError("prepare: Trees have duplicate leaves.")
    rooted_topo, _ = read_newick_file(rooted_tree_path)[0]                # :176
    og = min(root_bipartition(rooted_topo), key=lambda st: (len(st), sorted(st)))   # fst . fromBipartition
    rooted = [reroot_at_outgroup(t, ln, og) for t, ln in trs]             # :179-180
    sig0 = (tuple(rooted[0][0].parent), tuple(n if lf else "" for n, lf in zip(rooted[0][0].names, rooted[0][0].leaves)))
    for t, _ in rooted:                                                   # :184-193
        if (tuple(t.parent), tuple(n if lf else "" for n, lf in zip(t.names, t.leaves))) != sig0:
            raise TreeError("prepare: A single topology and equal sub tree orders are required.")
    if root_bipartition(rooted[0][0]) != root_bipartition(rooted_topo):   # :195-203 (necessary condition)
        raise TreeError("prepare: A single topology is required.")
    topo = rooted[0][0]
    pm_r = np.stack([sum_first_two(get_branches(topo, ln)) for _, ln in rooted])      # :103-104, :207
    mu = pm_r.mean(axis=0)                                                # meanCov, :208
    sigma = np.atleast_2d(np.cov(pm_r, rowvar=False, ddof=1))
    variances = np.diag(sigma)
    if variances.min() <= 0:                                              # :220
        raise ValueError("prepare: Minimum variance is zero or negative.")
    mean_lengths = np.stack([ln for _, ln in rooted]).mean(axis=0)        # :291-293
    if likelihood_spec == "FullMultivariateNormal":
        sign, logdet = np.linalg.slogdet(sigma)                           # invlndet, :230
        if sign != 1.0:                                                   # :231
            raise ValueError("prepare: Determinant of covariance matrix is negative?")
        lhd: LikelihoodData = Full(mu, np.linalg.inv(sigma), float(logdet))           # :240
    elif likelihood_spec == "UnivariateNormal":
        lhd = Univariate(mu, variances.copy())                            # :278-281
    elif likelihood_spec == "NoLikelihood":
        lhd = NoData()                                                    # :282-284
    elif likelihood_spec.startswith("SparseMultivariateNormal"):
        # app/Main.hs:257-276: centre and scale the columns (Statistics.Covariance.scale), graphical lasso with penalty rho on the
        # correlation matrix, scale covariance and precision back, ln det of the sparse covariance, association list
        parts = likelihood_spec.split()
        rho = float(parts[1]) if len(parts) > 1 else 0.1                  # scripts/run:137 passes 0.1
        sd = np.sqrt(variances)
        corr = sigma / np.outer(sd, sd)
        w_n, theta_n = graphical_lasso(corr, rho, penalize_diagonal=GLASSO_PENALIZE_DIAGONAL)
        sigma_s = w_n * np.outer(sd, sd)                                  # rescaleSWith
        prec_s = theta_n / np.outer(sd, sd)                               # rescalePWith
        sign, logdet = np.linalg.slogdet(sigma_s)
        if sign != 1.0:
            raise ValueError("prepare: Determinant of sparse covariance matrix is negative?")
        n = len(mu)
        assoc = [((i, j), float(prec_s[i, j])) for i in range(n) for j in range(n) if prec_s[i, j] != 0.0]
        lhd = Sparse(mu, assoc, float(logdet))
    else:
        raise NotImplementedError(f"prepare: likelihood specification {likelihood_spec!r} is not available")
    return Prepared(lhd, mu, sigma, topo, mean_lengths, n_trees, n_burn)


# The reference calls glasso (Friedman, Hastie, Tibshirani 2008; the Fortran code behind the Haskell packages `glasso` and
# `covariance`, neither vendored) with its default flags; whether its binding penalises the diagonal (R's wrapper does by
# default) is not visible from the reference's sources.  The posterior samples the reference commits for the mtCDNApri analysis
# decide (tests/test_gpu_mh.py::test_posterior_node_ages_against_the_references_own_samples, tools/post_samples_check.py).
GLASSO_PENALIZE_DIAGONAL = True


def graphical_lasso(S: np.ndarray, rho: float, penalize_diagonal: bool = True, tol: float = 1e-10, max_iter: int = 10000):
    """Graphical lasso: maximise ln det Theta - tr(S Theta) - rho ||Theta||_1 (the l1 norm over all entries, or over the
    off-diagonal ones).  Block coordinate descent over the columns of W = Theta^-1 with the lasso sub-problem solved by
    coordinate descent (Friedman, Hastie, Tibshirani, Biostatistics 9 (2008), section 2); the optimum is unique, so this
    agrees with any other solver to the tolerance.  Returns (W, Theta)."""
    S = np.asarray(S, float)
    p = S.shape[0]
    if rho < 0:
        raise ValueError("graphical_lasso: negative penalty")
    W = S.copy()
    if penalize_diagonal:
        W[np.diag_indices(p)] += rho
    if p == 1:
        return W, 1.0 / W
    B = np.zeros((p, p))                                                  # lasso coefficients per column
    idx = np.arange(p)
    for _ in range(max_iter):
        W_old = W.copy()
        for j in range(p):
            rest = idx != j
            W11 = W[np.ix_(rest, rest)]
            s12 = S[rest, j]
            beta = B[rest, j].copy()
            for _ in range(max_iter):                                     # coordinate descent on 1/2 b'W11 b - b's12 + rho |b|_1
                delta = 0.0
                for k in range(p - 1):
                    r = s12[k] - W11[k] @ beta + W11[k, k] * beta[k]
                    nb = np.sign(r) * max(abs(r) - rho, 0.0) / W11[k, k]
                    delta = max(delta, abs(nb - beta[k]))
                    beta[k] = nb
                if delta <= tol:
                    break
            B[rest, j] = beta
            w12 = W11 @ beta
            W[rest, j] = w12
            W[j, rest] = w12
        if np.abs(W - W_old).max() <= tol * max(1.0, np.abs(S - np.diag(np.diag(S))).mean()):
            break
    Theta = np.zeros((p, p))
    for j in range(p):
        rest = idx != j
        t22 = 1.0 / (W[j, j] - W[rest, j] @ B[rest, j])
        Theta[j, j] = t22
        Theta[rest, j] = -B[rest, j] * t22
    Theta = 0.5 * (Theta + Theta.T)
    Theta[np.abs(Theta) < 1e-14] = 0.0
    return W, Theta


def write_prepared(name: str, p: Prepared) -> None:
    """`<name>.data` and `<name>.meantree` (app/Main.hs:286, 305-307)."""
    write_data_file(name + ".data", p.lhd)
    with open(name + ".meantree", "w") as f:
        f.write(p.mean_tree_newick() + "\n")
