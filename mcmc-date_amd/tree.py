"""Host-side tree types mirroring the reference's `Mcmc.Tree.Types` / `Tools` for the hot path.

Only what the likelihood path needs: topology as a pre-order parent array (the order of
elynx-tree's `branches`, lib/Mcmc/Tree/Types.hs:91-95), Newick import, the canonical branch
order `getBranches` / `sumFirstTwo` (app/Tools.hs:36-48) and `heightTreeToLengthTree`
(lib/Mcmc/Tree/Types.hs:224-233).  These host functions serve setup and inspection; the per-step
evaluation of the same formulas happens inside the HIP kernels (csrc/mvn_kernels.hip, load_tree).
"""
from __future__ import annotations

import gzip
import re
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np


class TreeError(ValueError):
    """Structural fault; the reference calls `error` (e.g. app/Tools.hs:43)."""


@dataclass
class Topology:
    """Rooted tree topology, nodes numbered in pre-order (root = 0)."""

    parent: np.ndarray                      # int32 [n_nodes]; parent[0] = -1
    names: List[str] = field(default_factory=list)

    def __post_init__(self):
        self.parent = np.ascontiguousarray(self.parent, dtype=np.int32)
        if not self.names:
            self.names = [""] * len(self.parent)
        if len(self.parent) == 0 or self.parent[0] != -1:
            raise TreeError("Topology: parent[0] must be -1 (root)")
        stack = [0]
        for v in range(1, len(self.parent)):
            while stack and stack[-1] != self.parent[v]:
                stack.pop()
            if not stack:
                raise TreeError(f"Topology: node {v} is not numbered in pre-order")
            stack.append(v)

    @property
    def n_nodes(self) -> int:
        return len(self.parent)

    def children(self, v: int) -> List[int]:
        return [int(c) for c in np.nonzero(self.parent == v)[0]]

    @property
    def leaves(self) -> np.ndarray:
        is_leaf = np.ones(self.n_nodes, bool)
        is_leaf[self.parent[1:]] = False
        return is_leaf

    def subtree_size(self, v: int) -> int:
        e = v + 1
        while e < self.n_nodes:
            a = e
            while a > v:
                a = self.parent[a]
            if a != v:
                break
            e += 1
        return e - v

    def root_children(self):
        ch = self.children(0)
        if len(ch) != 2:
            raise TreeError("getBranches: Root node is not bifurcating.")  # app/Tools.hs:43
        return ch[0], ch[1]


def parse_newick(s: str):
    """Newick string -> (Topology, branch lengths per pre-order node)."""
    s = re.sub(r"\[[^\]]*\]", "", s).strip()      # Newick comments / annotations such as [&index=25]
    if not s.endswith(";"):
        raise TreeError("newick: missing ';'")
    pos = 0
    parent, length, names = [], [], []

    def node(par):
        nonlocal pos
        me = len(parent)
        parent.append(par)
        length.append(0.0)
        names.append("")
        while s[pos] == " ":
            pos += 1
        if s[pos] == "(":
            pos += 1
            while True:
                node(me)
                while s[pos] == " ":
                    pos += 1
                if s[pos] == ",":
                    pos += 1
                elif s[pos] == ")":
                    pos += 1
                    break
                else:
                    raise TreeError(f"newick: unexpected {s[pos]!r} at {pos}")
        while s[pos] == " ":
            pos += 1
        if s[pos] == "'":                         # quoted label, e.g. MCMCtree's 'B(6,8,2.5e-2,2.5e-2)': commas and brackets inside
            st = pos + 1
            pos = s.index("'", st)
            names[me] = s[st:pos]
            pos += 1
            while s[pos] == " ":
                pos += 1
        else:
            st = pos
            while s[pos] not in ":,();":
                pos += 1
            names[me] = s[st:pos].strip()
        if s[pos] == ":":
            pos += 1
            st = pos
            while s[pos] not in ",();":
                pos += 1
            length[me] = float(s[st:pos])

    node(-1)
    return Topology(np.asarray(parent, np.int32), names), np.asarray(length, np.float64)


def read_newick_file(path: str):
    """All trees of a Newick file (gz aware, cf. lib/Mcmc/Tree/Import.hs:61-76)."""
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        txt = re.sub(r"\[[^\]]*\]", "", f.read()).replace("\n", "")
    return [parse_newick(t + ";") for t in txt.split(";") if t.strip()]


def get_branches(topo: Topology, values: Sequence[float]) -> np.ndarray:
    """app/Tools.hs:36-43: [br l, br r] ++ tail (branches l) ++ tail (branches r); stem ignored."""
    v = np.asarray(values, dtype=np.float64)
    l, r = topo.root_children()
    sl, sr = topo.subtree_size(l), topo.subtree_size(r)
    return np.concatenate([[v[l], v[r]], v[l + 1:l + sl], v[r + 1:r + sr]])


def sum_first_two(v: np.ndarray) -> np.ndarray:
    """app/Tools.hs:47-48."""
    v = np.asarray(v, dtype=np.float64)
    return np.concatenate([[v[0] + v[1]], v[2:]])


def branch_slots(topo: Topology) -> np.ndarray:
    """Distance-vector slot of every node's branch (both root children -> 0, root -> -1)."""
    ids = get_branches(topo, np.arange(topo.n_nodes, dtype=np.float64)).astype(np.int64)
    slot = np.full(topo.n_nodes, -1, dtype=np.int64)
    for i, v in enumerate(ids):
        slot[v] = 0 if i < 2 else i - 1
    return slot


def height_tree_to_length_tree(topo: Topology, heights: np.ndarray) -> np.ndarray:
    """lib/Mcmc/Tree/Types.hs:224-233: l = hParent - hNode; the root gets hRoot - hRoot."""
    h = np.asarray(heights, dtype=np.float64)
    par = np.where(topo.parent >= 0, topo.parent, np.arange(topo.n_nodes))
    return h[..., par] - h


def is_valid_height_tree(topo: Topology, heights: np.ndarray) -> bool:
    """lib/Mcmc/Tree/Types.hs:181-185: leaves at 0, parents strictly above children."""
    h = np.asarray(heights, dtype=np.float64)
    if np.any(h[topo.leaves] != 0):
        return False
    return bool(np.all(h[topo.parent[1:]] > h[1:]))
