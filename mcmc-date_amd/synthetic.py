"""Synthetic workloads of BASELINE.json's configs: dense random SPD covariance,
random tree topologies and chain states.  Deterministic in the seed (numpy PCG64)."""
from __future__ import annotations

import numpy as np

from .state import StateBatch
from .tree import Topology


def random_spd_problem(n: int, seed: int):
    """Config 3 recipe: Sigma = (A A^T / n + 0.1 diag(u)) scaled to branch-length-like variances
    (1e-4 .. 1e-3), mu ~ U(0.01, 1).  Returns (mu, sigma)."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, n))
    u = rng.uniform(0.5, 1.5, n)
    S = A @ A.T / n + 0.1 * np.diag(u)
    sd = np.sqrt(rng.uniform(1e-4, 1e-3, n) / np.diag(S))
    S = S * np.outer(sd, sd)
    S = 0.5 * (S + S.T)
    mu = rng.uniform(0.01, 1.0, n)
    return mu, S


def sample_chains(mu, sigma, batch: int, seed: int) -> np.ndarray:
    """x_b = mu + L z_b, z ~ N(0, I): chain-major [batch, n]."""
    rng = np.random.default_rng(seed + 7919)
    L = np.linalg.cholesky(sigma)
    Z = rng.standard_normal((batch, len(mu)))
    return mu[None, :] + Z @ L.T


def random_topology(n_leaves: int, seed: int) -> Topology:
    """Random rooted binary tree with `n_leaves` leaves, pre-order numbering."""
    rng = np.random.default_rng(seed + 104729)

    def build(k):
        if k == 1:
            return None
        a = int(rng.integers(1, k))
        return (build(a), build(k - a))

    parent = []

    def number(t, par):
        me = len(parent)
        parent.append(par)
        if t is not None:
            number(t[0], me)
            number(t[1], me)

    number(build(n_leaves), -1)
    return Topology(np.asarray(parent, np.int32))


def random_states(topo: Topology, batch: int, seed: int, jitter: float = 0.05) -> StateBatch:
    """Valid states: ultrametric relative heights (root 1, leaves 0), log-normal rates around 1,
    tH and rMu log-normal around 1 (SURVEY.md 8d config 2 recipe)."""
    rng = np.random.default_rng(seed + 15485863)
    nn = topo.n_nodes
    leaves = topo.leaves
    H = np.zeros((batch, nn))
    # heights: root 1; every internal node = parent height * U(0.3, 0.9); leaves 0
    for v in range(nn):
        if v == 0:
            H[:, 0] = 1.0
        elif not leaves[v]:
            H[:, v] = H[:, topo.parent[v]] * rng.uniform(0.3, 0.9, batch)
    R = np.exp(jitter * 4 * rng.standard_normal((batch, nn)))
    R[:, 0] = 0.0
    tH = np.exp(jitter * rng.standard_normal(batch))
    rMu = np.exp(jitter * rng.standard_normal(batch))
    return StateBatch(H, R, tH, rMu)


def banded_precision(n: int, seed: int, band: int = 3, extra: int = 4, scale: float = 1e3):
    """A symmetric, strictly diagonally dominant (hence positive definite) precision matrix: a band plus `extra` random
    off-diagonal entries per row, values of mixed sign -- the density a graphical-lasso estimate of a large tree has
    (the reference's `Sparse` likelihood data, app/Probability.hs:178-184).  Returns (scipy CSR matrix, association list
    [((i, j), value)])."""
    import scipy.sparse as sps

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
    P = ((A + sps.diags(diag)) * scale).tocsr()              # precisions of branch lengths ~ 1e-2: like the fixtures
    coo = P.tocoo()
    return P, [((int(a), int(b)), float(c)) for a, b, c in zip(coo.row, coo.col, coo.data)]
