"""Batched Metropolis-Hastings-Green driver on the device (first slice).

Host-side mirror of what the reference assembles in app/Definitions.hs and hands to `mcmc`'s `mhg`
(app/Main.hs:460-479): the initial state (`initWith`, :96-123), the proposal cycle (`proposals`, :256-278 and the
helper lists :145-253), the burn-in schedule with auto tuning (`burnIn`, :420-424) and the iterations (:440-441).
`mcmc` evaluates one state per call; here B independent chains run in lock step on one GPU through `mcd_mh_*`
(include/mcmcdate_mvn.h): every chain executes the same proposal of the (shuffled) cycle at the same time with its
own random numbers, tuning parameter and accept/reject decision.  There is no CPU path.

Built: every proposal of the reference's Metropolis-Hastings cycle (`proposals bs calibrationsAvailable x Nothing`).
NUTS (`Just htarget`, app/Hamiltonian.hs) is not; `proposals()` returns the names of what it
leaves out (an empty list for the default cycle) so that nothing is skipped silently.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _capi
from .likelihood import TreeLikelihood
from .prior import PriorFunction
from .state import State, StateBatch
from .tree import Topology

SCALE_SCALAR, SLIDE_NODE, SCALE_SUBTREE_TIME, PULLEY, SCALE_BRANCH_RATE, SCALE_SUBTREE_RATE, SCALE_NORM_TREE, \
    SCALE_VAR_TREE, SCALE_VAR_TREE_AUTO, SCALE_CONTRARILY, SLIDE_NODE_CONTRA, SCALE_SUBTREE_CONTRA, SLIDE_ROOT_CONTRA, \
    SCALE_RATES_TREE_CONTRA, SLIDE_BRACE, SLIDE_BRACE_CONTRA = range(16)
BIRTH, DEATH, TIME_HEIGHT, RATE_MEAN, RATE_VARIANCE = range(5)


@dataclass
class Proposal:
    """One row of the proposal table (`Proposal I` in `mcmc`)."""
    name: str
    kind: int
    node: int = 0
    p0: float = 1.0          # standard deviation or gamma shape
    p1: float = 0.0          # second parameter (scaleContrarily: gamma scale; scaleVarianceAndTree: 1 = exact Jacobian)
    n1: int = 0
    n2: int = 0
    jac_root: int = 0        # liftProposalWith jacobianRootBranch (the "[R]" proposals): 1 = ratio jf(y) / jf(x) joins the acceptance ratio (-1: its reciprocal, experiments only)
    dim: int = 1             # PDimension
    weight: int = 1          # PWeight


# ---- initWith -- app/Definitions.hs:96-123 ------------------------------------------------------------------------
def init_with(topo: Topology, lengths: Sequence[float]) -> State:
    """Initial state from the mean tree: zero branches -> average branch, stem 0, terminal branches elongated until
    the tree is ultrametric, height normalised to 1, all rates 1 (stem 0), every scalar 1."""
    par = topo.parent
    n = topo.n_nodes
    ln = np.array(lengths, dtype=np.float64)
    avg = ln[1:].sum() / (n - 1)
    ln[1:][ln[1:] == 0] = avg
    ln[0] = 0.0
    dist = np.zeros(n)
    for v in range(1, n):
        dist[v] = dist[par[v]] + ln[v]
    height = dist[topo.leaves].max()
    heights = (height - dist) / height          # makeUltrametric, normalizeHeight, toHeightTreeUltrametric [elynx-tree]
    heights[topo.leaves] = 0.0
    heights[0] = 1.0
    rates = np.ones(n)
    rates[0] = 0.0
    return State(1.0, 1.0, 1.0, heights, 1.0, 1.0, rates)


# ---- proposals -- app/Definitions.hs:127-278 ----------------------------------------------------------------------
def weight_n_branches(n: int) -> int:
    """weightNBranches, :127-130."""
    return int(math.floor(math.log(n) / math.log(1.3)))


def _tables(topo: Topology):
    n = topo.n_nodes
    par = topo.parent
    size = np.ones(n, np.int64)
    inner = (~topo.leaves).astype(np.int64)      # nInnerNodes, Internal.hs:83-85
    levels = np.ones(n, np.int64)                # elynx `depth`: a leaf has depth 1
    plen = np.zeros(n, np.int64)                 # length of the path from the root
    for v in range(1, n):
        plen[v] = plen[par[v]] + 1
    for v in range(n - 1, 0, -1):
        size[par[v]] += size[v]
        inner[par[v]] += inner[v]
        levels[par[v]] = max(levels[par[v]], levels[v] + 1)
    return size, inner, levels, plen


def proposals(topo: Topology, braces: Sequence = (), calibrations_available: bool = False,
              exact_jacobians: bool = False) -> Tuple[List[Proposal], List[str]]:
    """The proposal cycle of `proposals bs calibrationsAvailable x Nothing` (:256-278) in the reference's order.
    Returns (table, names of reference proposals that are not built yet).

    exact_jacobians: two proposals of the reference use a Jacobian that is not the determinant of their map (DESIGN.md
    section 9): scaleVarianceAndTree takes the product of the diagonal, (u - u/n + 1/n)^n instead of u^(n-1), and
    slideRootContrarily divides by u once more than there are scaled heights.  False (default) restates the reference
    (parity: same stationary distribution as the reference, small bias included); True uses the determinants."""
    n = topo.n_nodes
    size, inner, levels, plen = _tables(topo)
    leaf = topo.leaves
    w = weight_n_branches(n)
    l, r = topo.root_children()
    ps: List[Proposal] = []
    missing: List[str] = []
    # :258-263
    ps.append(Proposal("Time birth rate", SCALE_SCALAR, BIRTH, 10.0, weight=w))
    ps.append(Proposal("Time death rate", SCALE_SCALAR, DEATH, 10.0, weight=w))
    ps.append(Proposal("Rate mean", SCALE_SCALAR, RATE_MEAN, 10.0, weight=w))
    ps.append(Proposal("Rate variance", SCALE_SCALAR, RATE_VARIANCE, 10.0, weight=w))
    n_inner = int(inner[0])
    if n_inner - 1 < 1:
        raise ValueError("scaleRatesAndTreeContrarilyPFunction: no internal nodes to scale")     # Contrary.hs:427
    ps.append(Proposal("Rates and time tree", SCALE_RATES_TREE_CONTRA, 0, 0.1, n1=n_inner - 1, jac_root=True, dim=n_inner - 1 + 2, weight=w))

    children_of_root = lambda v: plen[v] == 1      # :133-134
    other_nodes = lambda v: plen[v] > 1            # :137-138
    sub_w = lambda v: min(3 + int(levels[v]) - 2, 8)

    # proposalsTimeTree, :145-166
    def time_ps(hn, tag, jac):
        out = [Proposal(f"{tag} Time tree node {v}", SLIDE_NODE, v, 0.01, jac_root=jac, dim=1, weight=5)
               for v in range(n) if not leaf[v] and hn(v)]
        out += [Proposal(f"{tag} Time tree node {v}", SCALE_SUBTREE_TIME, v, 0.01, n1=int(inner[v]), jac_root=jac,
                         dim=int(inner[v]), weight=sub_w(v)) for v in range(n) if not leaf[v] and hn(v)]
        return out

    if not leaf[l] and not leaf[r]:
        ps.append(Proposal("[R] Time tree", PULLEY, 0, 0.01, n1=int(inner[l]), n2=int(inner[r]), jac_root=True,
                           dim=int(inner[l] + inner[r]), weight=6))
    ps += time_ps(children_of_root, "[R]", True)
    ps += time_ps(other_nodes, "[O]", False)
    for i, b in enumerate(braces):
        ps.append(Proposal(f"[B] Time tree {b.name}", SLIDE_BRACE, i, 0.01, dim=len(b.nodes), weight=5))

    # proposalsRateTree, :180-201
    ps.append(Proposal("[R] Rate mean, Rate tree", SCALE_NORM_TREE, RATE_MEAN, 100.0, jac_root=True, dim=n, weight=w))
    ps.append(Proposal("[R] Rate variance, Rate tree", SCALE_VAR_TREE, 0, 100.0, p1=1.0 if exact_jacobians else 0.0, jac_root=True, dim=n, weight=w))
    ps.append(Proposal("[R] Rate variance, Rate tree (autocorrelated)", SCALE_VAR_TREE_AUTO, 0, 100.0, jac_root=True, dim=n, weight=w))

    def rate_ps(hn, tag, jac):
        out = [Proposal(f"{tag} Rate tree branch {v}", SCALE_BRANCH_RATE, v, 100.0, jac_root=jac, dim=1, weight=3)
               for v in range(n) if hn(v)]
        out += [Proposal(f"{tag} Rate tree node {v}", SCALE_SUBTREE_RATE, v, 100.0, n1=int(size[v]), jac_root=jac,
                         dim=int(size[v]), weight=sub_w(v)) for v in range(n) if not leaf[v] and hn(v)]
        return out

    ps += rate_ps(children_of_root, "[R]", True)
    ps += rate_ps(other_nodes, "[O]", False)

    # proposalsTimeRateTreeContra, :204-221
    def contra_ps(hn, tag, jac):
        out = [Proposal(f"{tag} Trees node {v}", SLIDE_NODE_CONTRA, v, 0.1, jac_root=jac, dim=1 + 1 + len(topo.children(v)), weight=sub_w(v))
               for v in range(n) if not leaf[v] and hn(v)]
        out += [Proposal(f"{tag} Trees node {v}", SCALE_SUBTREE_CONTRA, v, 0.1, n1=int(inner[v]), n2=int(size[v]), jac_root=jac,
                         dim=int(inner[v] + size[v]), weight=sub_w(v)) for v in range(n) if not leaf[v] and hn(v)]
        return out

    ps += contra_ps(children_of_root, "[C] [R]", True)
    ps += contra_ps(other_nodes, "[C] [O]", False)
    for i, b in enumerate(braces):
        n_daughters = sum(len(topo.children(x)) for x in b.nodes)
        ps.append(Proposal(f"[C] [B] Trees {b.name}", SLIDE_BRACE_CONTRA, i, 0.1, dim=2 * len(b.nodes) + n_daughters, weight=5))

    # proposalsChangingTimeHeight, :241-253
    if calibrations_available:
        ps.append(Proposal("Time height", SCALE_SCALAR, TIME_HEIGHT, 3000.0, weight=w))
        ps.append(Proposal("Time height, rate mean", SCALE_CONTRARILY, 0, 10.0, 0.1, dim=2, weight=w))
        ps.append(Proposal("[R] Time height, Rate tree", SCALE_NORM_TREE, TIME_HEIGHT, 100.0, jac_root=True, dim=n, weight=w))
        ps.append(Proposal("[R] Trees", SLIDE_ROOT_CONTRA, 0, 10.0, n1=n_inner - 1 if exact_jacobians else n_inner, jac_root=True,
                           dim=1 + n_inner + 2, weight=w))
    return ps, missing


def table_arrays(ps: Sequence[Proposal]) -> dict:
    """Struct-of-arrays form of the table (what mcd_mh_create takes)."""
    i32 = lambda f: np.ascontiguousarray([int(f(p)) for p in ps], dtype=np.int32)
    f64 = lambda f: np.ascontiguousarray([float(f(p)) for p in ps], dtype=np.float64)
    return dict(kind=i32(lambda p: p.kind), node=i32(lambda p: p.node), n1=i32(lambda p: p.n1), n2=i32(lambda p: p.n2),
                jac_root=i32(lambda p: p.jac_root), dim=i32(lambda p: p.dim), p0=f64(lambda p: p.p0), p1=f64(lambda p: p.p1))


def cycle_schedule(ps: Sequence[Proposal], n_iter: int, rng: np.random.Generator) -> np.ndarray:
    """`mcmc`'s default cycle order [external]: every iteration executes each proposal `weight` times, in a freshly
    shuffled order.  [n_iter, sum of weights] int32."""
    base = np.repeat(np.arange(len(ps), dtype=np.int32), [p.weight for p in ps])
    out = np.empty((n_iter, len(base)), np.int32)
    for i in range(n_iter):
        out[i] = rng.permutation(base)
    return out


# burnIn -- app/Definitions.hs:420-424: tuning periods (all proposals built here are `PFast`)
BURN_IN_FAST = [10, 10] + list(range(10, 131, 10))
BURN_IN_SLOW = list(range(100, 401, 20))
ITERATIONS = 8000   # :440-441

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class Sampler:
    """B chains on one GPU.  `tree_lik` and `prior` must live on the same device and outlive the sampler."""

    def __init__(self, tree_lik: TreeLikelihood, prior: PriorFunction, table: Sequence[Proposal], batch: int, seed: int,
                 first_chain: int = 0):
        self.table = list(table)
        self.topo: Topology = tree_lik.topo
        self.batch = int(batch)
        self.seed = int(seed)
        self.first_chain = int(first_chain)        # global index of chain 0 (shards.shard_sampler): the random-stream id
        self._keep = (tree_lik, prior)
        self._h = C.c_void_p()
        a = table_arrays(self.table)
        ip = lambda x: x.ctypes.data_as(_ip)
        dp = lambda x: x.ctypes.data_as(_dp)
        L = _capi.lib()
        # a SparseTreeLikelihood (precision matrix kept sparse on the device: trees beyond 1024 branches) takes mcd_mh_create_sparse
        create = L.mcd_mh_create_sparse if type(tree_lik).__name__ == "SparseTreeLikelihood" else L.mcd_mh_create
        _capi.check(create(C.byref(self._h), tree_lik._t, prior._p, len(self.table), ip(a["kind"]), ip(a["node"]),
                           ip(a["n1"]), ip(a["n2"]), ip(a["jac_root"]), ip(a["dim"]), dp(a["p0"]), dp(a["p1"]),
                           self.batch, C.c_uint64(seed)))
        if first_chain:
            _capi.check(L.mcd_mh_set_chain_offset(self._h, int(first_chain)))
        self._sched_rng = np.random.default_rng([int(seed), 0x5EED])
        self.iterations_done = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _capi.lib().mcd_mh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state --------------------------------------------------------------------------------------------------
    def set_state(self, s: StateBatch):
        nn = self.topo.n_nodes
        f = lambda x: np.ascontiguousarray(x, dtype=np.float64)
        if s.time_birth_rate is None or s.time_death_rate is None or s.rate_variance is None:
            raise ValueError("set_state: the state batch lacks time_birth_rate / time_death_rate / rate_variance")
        arr = [f(s.time_birth_rate), f(s.time_death_rate), f(s.time_height), f(s.heights), f(s.rate_mean), f(s.rate_variance), f(s.rates)]
        if arr[3].shape != (self.batch, nn) or arr[6].shape != (self.batch, nn) or any(a.shape != (self.batch,) for a in (arr[0], arr[1], arr[2], arr[4], arr[5])):
            raise ValueError("set_state: inconsistent state shapes")
        _capi.check(_capi.lib().mcd_mh_set_state(self._h, *[a.ctypes.data_as(_dp) for a in arr], nn))

    def set_initial_state(self, x: State):
        """Every chain starts from the same state (the reference starts its single chain from `initWith`)."""
        self.set_state(StateBatch.from_states([x] * self.batch))

    def state(self) -> StateBatch:
        nn, B = self.topo.n_nodes, self.batch
        birth, death, tH, rMu, rVar = (np.empty(B) for _ in range(5))
        H, R = np.empty((B, nn)), np.empty((B, nn))
        _capi.check(_capi.lib().mcd_mh_get_state(self._h, *[a.ctypes.data_as(_dp) for a in (birth, death, tH, H, rMu, rVar, R)], nn))
        return StateBatch(H, R, tH, rMu, birth, death, rVar)

    def posterior(self) -> np.ndarray:
        """[B, 3]: ln prior, ln likelihood, ln jacobianRootBranch of the current states."""
        post = np.empty((self.batch, 3))
        _capi.check(_capi.lib().mcd_mh_get_posterior(self._h, post.ctypes.data_as(_dp)))
        return post

    # -- stepping -----------------------------------------------------------------------------------------------
    def run_schedule(self, schedule: np.ndarray, accumulate: bool = False, trace: bool = False):
        sched = np.ascontiguousarray(schedule, dtype=np.int32)
        if sched.ndim != 2:
            raise ValueError("run_schedule: schedule must be [n_iter, steps_per_iter]")
        n_iter, S = sched.shape
        ta = np.empty((n_iter * S, self.batch)) if trace else None
        tk = np.empty((n_iter * S, self.batch), np.int8) if trace else None
        _capi.check(_capi.lib().mcd_mh_run(self._h, sched.ctypes.data_as(_ip), n_iter, S, int(bool(accumulate)),
                                           ta.ctypes.data_as(_dp) if trace else None,
                                           tk.ctypes.data_as(C.POINTER(C.c_int8)) if trace else None))
        self.iterations_done += n_iter
        return (ta, tk) if trace else None

    def run(self, n_iter: int, accumulate: bool = False, chunk: int = 256):
        """n_iter iterations of the shuffled cycle."""
        done = 0
        while done < n_iter:
            k = min(chunk, n_iter - done)
            self.run_schedule(cycle_schedule(self.table, k, self._sched_rng), accumulate=accumulate)
            done += k

    def last_path(self) -> str:
        """Which launch structure the last run took (mcd_mh_last_path; PATHS)."""
        return PATHS.get(int(_capi.lib().mcd_mh_last_path(self._h)), "unknown")

    def last_dynamic_lds(self) -> int:
        """LDS bytes per workgroup of the persistent kernel the last run launched (mcd_mh_last_dynamic_lds)."""
        return int(_capi.lib().mcd_mh_last_dynamic_lds(self._h))

    def autotune(self):
        _capi.check(_capi.lib().mcd_mh_tune(self._h))

    def burn_in(self, fast: Sequence[int] = BURN_IN_FAST, slow: Sequence[int] = BURN_IN_SLOW):
        """BurnInWithCustomAutoTuning fast slow: run each period, then tune (mcmc [external])."""
        for period in list(fast) + list(slow):
            self.run(period)
            self.autotune()

    # -- diagnostics --------------------------------------------------------------------------------------------
    def tuning(self):
        """(tuning parameters [B, P], accepted [B, P], tried [B, P]) since the last autotune / reset."""
        B, P = self.batch, len(self.table)
        t = np.empty((B, P))
        a = np.empty((B, P), np.int32)
        n = np.empty((B, P), np.int32)
        _capi.check(_capi.lib().mcd_mh_get_tuning(self._h, t.ctypes.data_as(_dp), a.ctypes.data_as(_ip), n.ctypes.data_as(_ip)))
        return t, a, n

    def set_tuning(self, t: np.ndarray):
        t = np.ascontiguousarray(t, dtype=np.float64)
        if t.shape != (self.batch, len(self.table)):
            raise ValueError("set_tuning: expected [batch, n_prop]")
        _capi.check(_capi.lib().mcd_mh_set_tuning(self._h, t.ctypes.data_as(_dp)))

    def reset_counters(self):
        _capi.check(_capi.lib().mcd_mh_reset_counters(self._h))

    def set_temperatures(self, beta: np.ndarray):
        """Reciprocal temperatures in (0, 1] per chain: chain b accepts with (prior x likelihood)^beta[b] (MC3)."""
        beta = np.ascontiguousarray(beta, dtype=np.float64)
        if beta.shape != (self.batch,):
            raise ValueError("set_temperatures: expected [batch]")
        _capi.check(_capi.lib().mcd_mh_set_temperatures(self._h, beta.ctypes.data_as(_dp)))

    def age_sums(self):
        """(sum, sum of squares [B, n_nodes], n): running sums of the absolute node ages tH * h_v."""
        B, nn = self.batch, self.topo.n_nodes
        s, q = np.empty((B, nn)), np.empty((B, nn))
        n = C.c_int64(0)
        _capi.check(_capi.lib().mcd_mh_get_age_sums(self._h, s.ctypes.data_as(_dp), q.ctypes.data_as(_dp), C.byref(n)))
        return s, q, int(n.value)

    def reset_age_sums(self):
        _capi.check(_capi.lib().mcd_mh_reset_age_sums(self._h))

    def node_age_summary(self):
        """Posterior mean and variance of every node age pooled over chains and accumulated iterations, plus the
        standard error of the mean estimated from the spread of the per-chain means."""
        s, q, n = self.age_sums()
        if n == 0:
            raise ValueError("node_age_summary: nothing accumulated")
        per_chain = s / n
        mean = per_chain.mean(axis=0)
        var = q.sum(axis=0) / (n * self.batch) - mean * mean
        sem = per_chain.std(axis=0, ddof=1) / math.sqrt(self.batch) if self.batch > 1 else np.full_like(mean, np.nan)
        return mean, var, sem


# ---- Metropolis-coupled MCMC -- `mc3 (MC3Settings (NChains 4) (SwapPeriod 2) (NSwaps 3))`, app/Main.hs:476-478 -----------
MC3_STREAM_DOMAIN = 0x4D43335F53574150          # "MC3_SWAP": keeps the swap draws apart from the proposal draws of the same seed
PATHS = {0: "none", 1: "whole schedule in one launch, factor resident in LDS", 2: "whole schedule in one launch, two chains per workgroup, the factor streamed once per step",
         3: "two launches per lock step, the ln prior of the proposal beside its likelihood", 4: "two launches per lock step (prior inside the step kernel)",
         5: "two launches per lock step: workgroup-per-chain step kernel leaving distances + plain-vector likelihood",
         6: "workgroup-per-chain step kernel + a likelihood launch only for proposals that move many distances (the others: columns of L^-1 on the kept z)",
         7: "two launches per lock step: workgroup-per-chain step kernel leaving distances + the sparse product (precision matrix in CSR)",
         8: "segments: the steps between two dense proposals in one launch with the chains' states in LDS; a dense proposal is proposed by the segment before it and evaluated by a likelihood launch",
         9: "segments over a sparse precision matrix: the steps between two dense proposals in one launch, the quadratic form updated through the rows of the "
            "moved distances; a dense proposal is proposed by the segment before it and evaluated by the one-launch full form"}


def philox4x32(counter, key):
    """Philox4x32-10 (Salmon et al. 2011), the generator of csrc/mh_device.hpp: philox_block, restated on Python integers for the
    host-side mirror of the swap phase (a dozen draws per group and phase)."""
    c0, c1, c2, c3 = (int(x) & 0xFFFFFFFF for x in counter)
    k0, k1 = (int(x) & 0xFFFFFFFF for x in key)
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0, c1, c2, c3


def uniform_pair(seed: int, chain: int, step: int, draw: int):
    """The two uniforms of draw `draw` in the stream (seed, chain, step): counter = (draw, chain, step lo, step hi), key = seed."""
    x = philox4x32((draw, chain, step & 0xFFFFFFFF, (step >> 32) & 0xFFFFFFFF), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    return ((((x[0] << 32) | x[1]) >> 11) + 0.5) * 2.0 ** -53, ((((x[2] << 32) | x[3]) >> 11) + 0.5) * 2.0 ** -53


def mc3_swap_host(rank: np.ndarray, lnpi: np.ndarray, ladder: np.ndarray, n_swaps: int, seed: int, phase: int, tried=None, accepted=None):
    """One swap phase over ALL global chains on the host -- the arithmetic of csrc/k_mc3.hip: k_mc3_swap on the same counter-based
    draws (used where no device is involved: the CPU rehearsals of the sharded loop, and as the check of the device kernel).
    rank [total] int (modified in place), lnpi [total] = ln prior + ln likelihood."""
    n = len(ladder)
    for g in range(len(rank) // n):
        base = g * n
        rk = rank[base:base + n]
        at = np.empty(n, np.int64)
        at[rk] = np.arange(n)
        pairs = list(range(n - 1))
        for j in range(n_swaps):
            ua, ub = uniform_pair(seed, g, phase, j)
            idx = min(j + int(ua * (n - 1 - j)), n - 2)
            i = pairs[idx]
            pairs[idx] = pairs[j]
            pairs[j] = i
            a, c = at[i], at[i + 1]
            log_r = (ladder[i] - ladder[i + 1]) * (lnpi[base + c] - lnpi[base + a])
            if tried is not None:
                tried[i] += 1
            if math.log(ub) < log_r:                                    # NaN compares false: no swap
                rk[a], rk[c] = i + 1, i
                at[i], at[i + 1] = c, a
                if accepted is not None:
                    accepted[i] += 1
    return rank


class MC3:
    """Metropolis-coupled MCMC (Geyer 1991; Altekar et al. 2004) over the lock-step driver: the GLOBAL set of chains is cut into
    groups of `n_chains` consecutive chains with reciprocal temperatures `betas` (rank 0 = cold, beta = 1); every `swap_period`
    iterations `n_swaps` distinct adjacent temperature pairs per group, in random order, propose to swap, with probability
    min(1, exp((beta_i - beta_j) (ln pi(x_j) - ln pi(x_i)))), pi = prior x likelihood.  Temperatures move between chains
    (the states stay where they are), which is the same Markov chain as swapping states.  Only cold chains are
    reported, like the reference's monitors.  The algorithm lives in the package `mcmc` [external, not vendored]: its
    initial ladder and its tuning of the ladder are not restated (parity unpinned); the default ladder 0.97^i is this
    build's choice.

    `backend`: a `Sampler` -- the swap phase then runs ON THE DEVICE behind the C ABI (mcd_mh_mc3_init / mcd_mh_mc3_swap,
    csrc/k_mc3.hip) -- or anything with run / posterior / set_temperatures / state / batch (the CPU twin in the tests), for
    which the same phase is evaluated by `mc3_swap_host` on the same counter-based draws.  A sharded run (one process per GPU):
    `shard` = this rank's ChainShard and `gather` = a callable that all-gathers the ranks' [3][batch] ln posterior arrays into
    [world][3][batch] (shards.ShardComm / shards.gather_posterior); every rank evaluates all groups, so the run does not depend
    on the number of ranks."""

    def __init__(self, backend, n_chains: int = 4, swap_period: int = 2, n_swaps: int = 3, betas: Optional[Sequence[float]] = None,
                 seed: int = 0, shard=None, gather=None):
        self.shard = shard
        total = backend.batch if shard is None else shard.n_chains
        if shard is not None and (shard.n_chains % shard.world != 0 or shard.size != backend.batch):
            raise ValueError("MC3: a sharded run needs equally large shards that match the backend's batch")
        if n_chains < 2 or total % n_chains != 0:
            raise ValueError("MC3: the number of chains must be a multiple of n_chains >= 2")
        if swap_period < 1 or not (1 <= n_swaps <= n_chains - 1):
            raise ValueError("MC3: need swap_period >= 1 and 1 <= n_swaps <= n_chains - 1")       # mcmc's own checks
        self.backend, self.n, self.period, self.n_swaps = backend, int(n_chains), int(swap_period), int(n_swaps)
        self.ladder = np.asarray(betas if betas is not None else [0.97 ** i for i in range(n_chains)], dtype=np.float64)
        if self.ladder.shape != (n_chains,) or self.ladder[0] != 1.0 or np.any(np.diff(self.ladder) >= 0) or np.any(self.ladder <= 0):
            raise ValueError("MC3: betas must start at 1 and decrease")
        self.total = int(total)
        self.lo = 0 if shard is None else shard.lo
        self.gather = gather
        if shard is not None and shard.world > 1 and gather is None:
            raise ValueError("MC3: a sharded run needs `gather`")
        self.seed = int(seed) ^ MC3_STREAM_DOMAIN
        self.phase = 0
        self.device = isinstance(backend, Sampler)
        if self.device:
            _capi.check(_capi.lib().mcd_mh_mc3_init(backend._h, self.n, self.ladder.ctypes.data_as(_dp), self.total, C.c_uint64(self.seed)))
        else:
            self._rank = (np.arange(self.total) % self.n).astype(np.int32)
            self._tried = np.zeros(n_chains - 1, np.int64)
            self._accepted = np.zeros(n_chains - 1, np.int64)
            self.backend.set_temperatures(self.ladder[self._rank[self.lo:self.lo + backend.batch]])

    # -- what the device holds ------------------------------------------------------------------------------------------
    def _get(self):
        rank = np.empty(self.total, np.int32)
        tried = np.empty(self.n - 1, np.int64)
        acc = np.empty(self.n - 1, np.int64)
        _capi.check(_capi.lib().mcd_mh_mc3_get(self.backend._h, rank.ctypes.data_as(_ip), tried.ctypes.data_as(C.POINTER(C.c_int64)),
                                               acc.ctypes.data_as(C.POINTER(C.c_int64)), None))
        return rank, tried, acc

    @property
    def rank(self) -> np.ndarray:
        """Temperature rank of every GLOBAL chain."""
        return self._get()[0] if self.device else self._rank

    @property
    def swaps_tried(self) -> np.ndarray:
        return self._get()[1] if self.device else self._tried

    @property
    def swaps_accepted(self) -> np.ndarray:
        return self._get()[2] if self.device else self._accepted

    def cold(self) -> np.ndarray:
        """Local indices of this backend's chains that are cold right now."""
        return np.nonzero(self.rank[self.lo:self.lo + self.backend.batch] == 0)[0]

    def swap(self):
        """One swap phase."""
        world = 1 if self.shard is None else self.shard.world
        if self.device:
            if world == 1:
                _capi.check(_capi.lib().mcd_mh_mc3_swap(self.backend._h, self.n_swaps, None, 1, self.backend.batch))
            else:
                g = self.gather(self.backend)                          # device tensor [world][3][batch], on the sampler's stream
                self._keep_gathered = g
                _capi.check(_capi.lib().mcd_mh_mc3_swap(self.backend._h, self.n_swaps, C.c_void_p(g.data_ptr()), world, self.backend.batch))
        else:
            post = np.asarray(self.backend.posterior() if hasattr(self.backend, "posterior") else self.backend.post)   # [batch, 3]
            local = np.ascontiguousarray(post[:, :3].T)                # [3][batch]
            allp = local[None] if world == 1 else np.asarray(self.gather(local))
            lnpi = (allp[:, 0, :] + allp[:, 1, :]).reshape(-1)
            mc3_swap_host(self._rank, lnpi, self.ladder, self.n_swaps, self.seed, self.phase, self._tried, self._accepted)
            self.backend.set_temperatures(self.ladder[self._rank[self.lo:self.lo + self.backend.batch]])
        self.phase += 1

    def run(self, n_iter: int, collect_ages: bool = False):
        """n_iter iterations with a swap phase every swap_period iterations.  collect_ages: returns the absolute node ages
        tH * h_v of the cold chains after every swap phase, [n_phases, cold chains of this backend, n_nodes]."""
        out = []
        done = 0
        while done < n_iter:
            k = min(self.period, n_iter - done)
            self.backend.run(k)
            done += k
            if k == self.period:
                self.swap()
            if collect_ages:
                s = self.backend.state()
                c = self.cold()
                out.append(np.asarray(s.time_height)[c, None] * np.asarray(s.heights)[c])
        return np.array(out) if collect_ages else None
