"""numpy restatement of McmcDate's `prepare` step and initial state -- TEST INFRASTRUCTURE ONLY.

Follows app/Main.hs:159-307 (`prepare`), app/Main.hs:101-104
(`getPosteriorMatrixMergeBranchesToRoot`), app/Tools.hs:36-48 and app/Definitions.hs:96-123
(`initWith`).  Used by tests/golden/make_fixtures.py to turn the reference's own test inputs
(tests/<NN>-leaves-*/data/test.treelist) into committed fixtures (mu, Sigma, Sigma^-1, logdet,
sample vectors, expected log-likelihoods).  PARITY UNPINNED: the reference commits no outputs
of `prepare`; hmatrix's meanCov / invlndet ([ext], LAPACK) are restated with numpy.

Trees are held as pre-order arrays (see oracle/mvn_oracle.c header).
"""
from __future__ import annotations

import gzip
import re
from dataclasses import dataclass

import numpy as np


@dataclass
class PTree:
    parent: np.ndarray   # int32 [n_nodes], pre-order, parent[0] = -1
    length: np.ndarray   # float64 [n_nodes] branch length above each node (root: stem)
    name: list           # str per node ('' for unnamed)

    @property
    def n_nodes(self):
        return len(self.parent)

    def children(self, v):
        return [int(c) for c in np.nonzero(self.parent == v)[0]]

    def is_leaf(self, v):
        return not np.any(self.parent == v)

    def leaves_below(self, v):
        size = subtree_size(self.parent, v)
        return frozenset(self.name[u] for u in range(v, v + size) if self.is_leaf(u))


def subtree_size(parent, v):
    n = len(parent)
    e = v + 1
    while e < n:
        a = e
        while a > v:
            a = parent[a]
        if a != v:
            break
        e += 1
    return e - v


def parse_newick(s: str) -> PTree:
    """Minimal Newick reader (names, branch lengths, no comments/quotes)."""
    s = re.sub(r"\[[^\]]*\]", "", s).strip()      # drop Newick comments / annotations
    assert s.endswith(";"), "newick string must end with ';'"
    pos = 0
    parent, length, name = [], [], []

    def node(par):
        nonlocal pos
        me = len(parent)
        parent.append(par); length.append(0.0); name.append("")
        if s[pos] == "(":
            pos += 1
            while True:
                node(me)
                if s[pos] == ",":
                    pos += 1
                    continue
                if s[pos] == ")":
                    pos += 1
                    break
                raise ValueError(f"newick: unexpected {s[pos]!r} at {pos}")
        st = pos
        while s[pos] not in ":,();":
            pos += 1
        name[me] = s[st:pos].strip()
        if s[pos] == ":":
            pos += 1
            st = pos
            while s[pos] not in ",();":
                pos += 1
            length[me] = float(s[st:pos])
        return me

    node(-1)
    assert s[pos] == ";"
    return PTree(np.asarray(parent, np.int32), np.asarray(length, np.float64), name)


def read_trees(path: str):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        txt = f.read()
    txt = re.sub(r"\[[^\]]*\]", "", txt)
    return [parse_newick(t + ";") for t in txt.replace("\n", "").split(";") if t.strip()]


def get_branches(t: PTree, values=None) -> np.ndarray:
    """app/Tools.hs:36-43."""
    v = t.length if values is None else np.asarray(values)
    ch = t.children(0)
    if len(ch) != 2:
        raise ValueError("getBranches: Root node is not bifurcating.")
    l, r = ch
    sl, sr = subtree_size(t.parent, l), subtree_size(t.parent, r)
    return np.concatenate([[v[l], v[r]], v[l + 1:l + sl], v[r + 1:r + sr]])


def sum_first_two(v: np.ndarray) -> np.ndarray:
    """app/Tools.hs:47-48."""
    return np.concatenate([[v[0] + v[1]], v[2:]])


def root_bipartition(t: PTree):
    l, r = t.children(0)
    return frozenset([t.leaves_below(l), t.leaves_below(r)])


def topology_signature(t: PTree):
    """Topology including sub-tree order (app/Main.hs:184-193 requires both to match)."""
    return (tuple(int(p) for p in t.parent), tuple(n if t.is_leaf(i) else "" for i, n in enumerate(t.name)))


@dataclass
class Prepared:
    mu: np.ndarray
    sigma: np.ndarray
    sigma_inv: np.ndarray
    logdet: float
    parent: np.ndarray          # topology shared by all trees (pre-order)
    names: list
    mean_lengths: np.ndarray    # per node, both root branches separately (mean tree, app/Main.hs:288-300)
    samples: np.ndarray         # [n_trees_kept, N] rows of the posterior matrix pmR


def prepare(treelist_path: str, rooted_tree_path: str) -> Prepared:
    trees_all = read_trees(treelist_path)                       # app/Main.hs:162
    n_burn = len(trees_all) // 6                                # :166
    trs = trees_all[n_burn:]                                    # :168
    rooted = read_trees(rooted_tree_path)[0]                    # :176
    # :179-180 `outgroup og`: the reference re-roots every tree at the rooted tree's outgroup.
    # All treelists shipped in tests/ are already rooted at that bipartition (PhyloBayes output,
    # one zero-length root child), so re-rooting is the identity up to how the root branch is
    # split -- which sumFirstTwo erases.  Anything else is refused rather than guessed.
    if root_bipartition(trs[0]) != root_bipartition(rooted):
        raise NotImplementedError("prepare: tree list is not rooted at the rooted tree's outgroup")
    sig = topology_signature(trs[0])                            # :184-193
    for t in trs:
        if topology_signature(t) != sig:
            raise ValueError("prepare: A single topology and equal sub tree orders are required.")
    pmR = np.stack([sum_first_two(get_branches(t)) for t in trs])     # :103-104, :207
    mu = pmR.mean(axis=0)                                             # :208 meanCov
    sigma = np.cov(pmR, rowvar=False, ddof=1)                         # (n-1)-normalised [ext hmatrix]
    if np.min(np.diag(sigma)) <= 0:                                   # :220
        raise ValueError("prepare: Minimum variance is zero or negative.")
    sign, logdet = np.linalg.slogdet(sigma)                           # :230 invlndet (LU)
    sigma_inv = np.linalg.inv(sigma)
    if sign != 1.0:                                                   # :231
        raise ValueError("prepare: Determinant of covariance matrix is negative?")
    pm = np.stack([t.length for t in trs])                            # getPosteriorMatrix, :291
    mean_lengths = pm.mean(axis=0)
    return Prepared(mu, sigma, sigma_inv, float(logdet), trs[0].parent.copy(), list(trs[0].name),
                    mean_lengths, pmR)


def node_heights_ultrametric(parent, length):
    """Heights of an ultrametric tree with leaves at 0 (toHeightTreeUltrametric, Types.hs:199-221)."""
    n = len(parent)
    depth = np.zeros(n)
    for v in range(1, n):
        depth[v] = depth[parent[v]] + length[v]
    is_leaf = np.ones(n, bool)
    is_leaf[parent[1:]] = False
    total = depth[is_leaf].max()
    h = total - depth
    h[is_leaf] = 0.0
    return h


def init_state(parent, mean_lengths):
    """initWith -- app/Definitions.hs:96-123: time tree = mean tree with zero branches replaced by
    the average, stem 0, terminal branches elongated to make it ultrametric, height normalised to
    1; all rates 1; tH = rMu = 1 (birth/death rates and rate variance do not enter the likelihood)."""
    parent = np.asarray(parent)
    n = len(parent)
    ln = np.asarray(mean_lengths, float).copy()
    bs = ln[1:]
    avg = bs.sum() / len(bs)                                    # :113-115
    ln[1:] = np.where(bs == 0, avg, bs)                         # :117
    ln[0] = 0.0                                                 # :119
    depth = np.zeros(n)
    for v in range(1, n):
        depth[v] = depth[parent[v]] + ln[v]
    is_leaf = np.ones(n, bool)
    is_leaf[parent[1:]] = False
    hmax = depth[is_leaf].max()
    ln[is_leaf] += hmax - depth[is_leaf]                        # makeUltrametric [ext elynx]
    ln /= hmax                                                  # normalizeHeight [ext elynx]
    heights = node_heights_ultrametric(parent, ln)
    heights[0] = 1.0
    rates = np.ones(n)
    rates[0] = 0.0                                              # setStem 0, :105
    return dict(heights=heights, rates=rates, tH=1.0, rMu=1.0)


# numpy twins of the C oracle (independent second implementation used when making fixtures)
def logpdf_full_np(mu, sigma_inv, logdet, x):
    dx = np.asarray(x) - mu
    return -0.9189385332046727 * len(mu) + (-0.5) * (logdet + (dx @ sigma_inv) @ dx)


def distances_np(parent, heights, rates, tH, rMu):
    parent = np.asarray(parent)
    t = PTree(parent, np.zeros(len(parent)), [""] * len(parent))
    ln = np.where(parent >= 0, heights[np.maximum(parent, 0)] - heights, 0.0)
    return sum_first_two(get_branches(t, ln) * get_branches(t, rates)) * (tH * rMu)
