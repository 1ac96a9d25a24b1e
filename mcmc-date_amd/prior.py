"""Host-side mirror of the reference's prior plugin on top of the C ABI.

`prior_function ht md cb cs bs` mirrors `priorFunction` (app/Probability.hs:127-150); `Calibration`,
`Constraint`, `Brace` and their loaders mirror lib/Mcmc/Tree/Prior/Node/{Calibration,Constraint,Brace}.hs
(file formats: Calibration.hs:286-319, Constraint.hs:306-374, Brace.hs:173-192), with nodes identified as
the most recent common ancestor of two leaves and stored as pre-order indices.  Evaluation happens on the
GPU (csrc/k_prior.hip); there is no CPU evaluation path here.
"""
from __future__ import annotations

import csv
import ctypes as C
import json
import re
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _capi
from .likelihood import _check_cuda, _is_torch, _ptr, _stream_ptr
from .state import State, StateBatch
from .tree import Topology, TreeError

CLOCK_MODELS = {"UncorrelatedGamma": 0, "UncorrelatedLogNormal": 1, "UncorrelatedWhiteNoise": 2, "AutocorrelatedLogNormal": 3}


@dataclass
class Calibration:
    """Node height calibration: interval in absolute time with soft boundaries (Calibration.hs:108-121)."""
    name: str
    node: int
    lower: Optional[float]           # None = Zero
    lower_p: float
    upper: Optional[float]           # None = Infinity
    upper_p: float


@dataclass
class Constraint:
    """Soft node order constraint: `young` is younger than `old` (Constraint.hs)."""
    name: str
    young: int
    old: int
    p: float


@dataclass
class Brace:
    """Soft brace: the listed nodes have (nearly) the same height (Brace.hs)."""
    name: str
    nodes: List[int]
    sd: float


def mrca(topo: Topology, leaf_a: str, leaf_b: str) -> int:
    """Pre-order index of the most recent common ancestor of two named leaves."""
    try:
        a, b = topo.names.index(leaf_a), topo.names.index(leaf_b)
    except ValueError as e:
        raise TreeError(f"mrca: leaf not found: {e}") from None
    anc = set()
    v = a
    while v >= 0:
        anc.add(v)
        v = int(topo.parent[v])
    v = b
    while v not in anc:
        v = int(topo.parent[v])
    return v


def _fnum(x: str) -> Optional[float]:
    x = x.strip()
    return float(x) if x else None


def load_calibrations(topo: Topology, path: str) -> List[Calibration]:
    """CSV rows `Name,LeafA,LeafB,LowerBoundary,LowerProbabilityMass,UpperBoundary,UpperProbabilityMass`;
    either boundary with its mass may be empty (Calibration.hs:286-319)."""
    out = []
    with open(path, newline="") as f:
        rows = list(csv.reader(f))[1:]
    if not rows:
        raise ValueError(f"loadCalibrations: No calibrations found in file: {path}.")
    for r in rows:
        if not r:
            continue
        lo, lop, hi, hip = _fnum(r[3]), _fnum(r[4]), _fnum(r[5]), _fnum(r[6])
        for p in (lop if lo is not None else None, hip if hi is not None else None):
            if p is not None and not (0 < p < 1):
                raise ValueError("probabilityMass: Zero or negative, or 1.0 or larger.")
        out.append(Calibration(r[0], mrca(topo, r[1], r[2]), lo, lop or 0.0, hi, hip or 0.0))
    nodes = [c.node for c in out]
    if len(set(nodes)) != len(nodes):
        raise ValueError("loadCalibrations: Duplicate/conflicting/redundant calibrations have been detected.")
    return out


_MCMCTREE_BOUND = re.compile(r"^\s*([LUB])\(([^()]*)\)\s*$")


def load_calibrations_from_tree(topo: Topology, path: str) -> List[Calibration]:
    """`loadCalibrationsFromTree` -- lib/Mcmc/Tree/Prior/Node/CalibrationFromTree.hs:119-130: calibrations given as node
    labels of a Newick tree in MCMCtree's notation (only L, U and B; PAML manual p. 49):
      'L(lower[, cauchyP, cauchyC[, p]])'  lower bound (the two Cauchy parameters are ignored, :29-46),
      'U(upper[, p])'                      upper bound (:49-63),
      'B(lower, upper[, pLower[, pUpper]])' both (:66-87);
    a missing probability mass defaults to 0.01 (`defPM`, :94-97).  A labelled node is identified by its leftmost and its
    rightmost leaf (`filterBoundedNodes`, :111-123) and looked up in the analysis' own rooted tree `topo` as their most
    recent common ancestor (`checkAndConvertCalibrationData`); its name is "leafA-leafB"."""
    from .tree import read_newick_file

    ctopo, _ = read_newick_file(path)[0]
    kids = [[] for _ in range(ctopo.n_nodes)]
    for v in range(1, ctopo.n_nodes):
        kids[int(ctopo.parent[v])].append(v)

    def edge_leaf(v: int, last: bool) -> str:
        while kids[v]:
            v = kids[v][-1 if last else 0]
        return ctopo.names[v]

    out = []
    for v in range(ctopo.n_nodes):
        m = _MCMCTREE_BOUND.match(ctopo.names[v])
        if not m:                                        # any other label (leaf names included) is not a calibration
            continue
        try:
            nums = [float(x) for x in m.group(2).split(",")]
        except ValueError:
            continue
        kind = m.group(1)
        lo = lop = hi = hip = None
        if kind == "L" and 1 <= len(nums) <= 4:
            lo, lop = nums[0], (nums[3] if len(nums) == 4 else None)
        elif kind == "U" and 1 <= len(nums) <= 2:
            hi, hip = nums[0], (nums[1] if len(nums) == 2 else None)
        elif kind == "B" and 2 <= len(nums) <= 4:
            lo, hi = nums[0], nums[1]
            lop = nums[2] if len(nums) >= 3 else None
            hip = nums[3] if len(nums) == 4 else None
        else:
            continue
        lop = 0.01 if (lo is not None and lop is None) else lop          # defPM
        hip = 0.01 if (hi is not None and hip is None) else hip
        for p in (lop, hip):
            if p is not None and not (0 < p < 1):
                raise ValueError("probabilityMass: Zero or negative, or 1.0 or larger.")
        a, b = edge_leaf(v, False), edge_leaf(v, True)
        out.append(Calibration(f"{a}-{b}", mrca(topo, a, b), lo, lop or 0.0, hi, hip or 0.0))
    if not out:
        raise ValueError(f"loadCalibrationsFromTree: no calibrations found in file: {path}")
    nodes = [c.node for c in out]
    if len(set(nodes)) != len(nodes):
        raise ValueError("loadCalibrations: Duplicate/conflicting/redundant calibrations have been detected.")
    return out


def load_constraints(topo: Topology, path: str) -> List[Constraint]:
    """CSV rows `Name,YoungerLeafA,YoungerLeafB,OlderLeafA,OlderLeafB,ProbabilityMass` (Constraint.hs:306-374)."""
    out = []
    with open(path, newline="") as f:
        rows = list(csv.reader(f))[1:]
    for r in rows:
        if not r:
            continue
        y, o = mrca(topo, r[1], r[2]), mrca(topo, r[3], r[4])
        if y == o:
            raise ValueError(f"loadConstraints: constraint {r[0]}: both nodes are equal")
        out.append(Constraint(r[0], y, o, float(r[5])))
    return out


def load_braces(topo: Topology, path: str) -> List[Brace]:
    """JSON list of {braceDataName, braceDataNodes: [[leafA, leafB], ...], braceDataStandardDeviation} (Brace.hs:173-192)."""
    out = []
    for b in json.load(open(path)):
        nodes = [mrca(topo, a, c) for a, c in b["braceDataNodes"]]
        if len(nodes) < 2:
            raise ValueError("loadBraces: a brace needs at least two nodes")
        out.append(Brace(b["braceDataName"], nodes, float(b["braceDataStandardDeviation"])))
    return out


def get_mean_root_height(cals: Sequence[Calibration]) -> Optional[float]:
    """`getMeanRootHeight` -- Calibration.hs:324-339: mean of the root calibration interval, if there is exactly one."""
    root = [c for c in cals if c.node == 0]
    if len(root) != 1 or root[0].upper is None:
        return None
    c = root[0]
    return c.upper / 2.0 if c.lower is None else (c.lower + c.upper) / 2.0


class PriorFunction:
    """`priorFunction ht md cb cs bs` with its tables staged once on one GPU."""

    def __init__(self, ht: float, model: str, calibrations: Sequence[Calibration], constraints: Sequence[Constraint],
                 braces: Sequence[Brace], topo: Topology, device: int = 0):
        if model not in CLOCK_MODELS:
            raise ValueError(f"unknown relaxed molecular clock model {model!r}")
        self.topo, self.device = topo, int(device)
        self._p = C.c_void_p()
        ia = lambda xs: np.ascontiguousarray(list(xs), dtype=np.int32)
        da = lambda xs: np.ascontiguousarray(list(xs), dtype=np.float64)
        cal = list(calibrations)
        con = list(constraints)
        br = list(braces)
        arrs = [ia(c.node for c in cal), ia(c.lower is not None for c in cal), da(c.lower or 0.0 for c in cal),
                da(c.lower_p for c in cal), ia(c.upper is not None for c in cal), da(c.upper or 0.0 for c in cal),
                da(c.upper_p for c in cal),
                ia(k.young for k in con), ia(k.old for k in con), da(k.p for k in con),
                np.concatenate([[0], np.cumsum([len(b.nodes) for b in br])]).astype(np.int32),
                ia(n for b in br for n in b.nodes), da(b.sd for b in br)]
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        par = np.ascontiguousarray(topo.parent, dtype=np.int32)
        _capi.check(_capi.lib().mcd_prior_create(
            C.byref(self._p), len(par), ip(par), C.c_double(ht), CLOCK_MODELS[model],
            len(cal), ip(arrs[0]), ip(arrs[1]), dp(arrs[2]), dp(arrs[3]), ip(arrs[4]), dp(arrs[5]), dp(arrs[6]),
            len(con), ip(arrs[7]), ip(arrs[8]), dp(arrs[9]),
            len(br), ip(arrs[10]), ip(arrs[11]), dp(arrs[12]), self.device))

    def close(self):
        if getattr(self, "_p", None) is not None and self._p.value:
            _capi.lib().mcd_prior_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def logprior(self, s: StateBatch, want_components: bool = False):
        """ln prior per chain (and, optionally, [batch, 3]: node priors, birth-death block, clock block)."""
        if s.time_birth_rate is None or s.time_death_rate is None or s.rate_variance is None:
            raise ValueError("logprior: the state batch lacks time_birth_rate / time_death_rate / rate_variance")
        L = _capi.lib()
        nn = self.topo.n_nodes
        fields = (s.time_birth_rate, s.time_death_rate, s.time_height, s.heights, s.rate_mean, s.rate_variance, s.rates)
        if _is_torch(s.heights):
            import torch

            for t in fields:
                _check_cuda(t, self.device, "state field")
            B = s.heights.shape[0]
            lp = torch.empty(B, dtype=torch.float64, device=s.heights.device)
            comp = torch.empty(B, 3, dtype=torch.float64, device=s.heights.device) if want_components else None
            _capi.check(L.mcd_prior_logprior_batch(self._p, *[_ptr(t) for t in fields[:4]], _ptr(fields[4]), _ptr(fields[5]),
                                                   _ptr(fields[6]), s.heights.stride(0), B, 1, _stream_ptr(self.device),
                                                   _ptr(lp), _ptr(comp) if want_components else None))
            return (lp, comp) if want_components else lp
        arr = [np.ascontiguousarray(a, dtype=np.float64) for a in fields]
        B = arr[3].shape[0]
        if arr[3].shape != (B, nn) or arr[6].shape != (B, nn) or any(a.shape != (B,) for a in (arr[0], arr[1], arr[2], arr[4], arr[5])):
            raise ValueError("logprior: inconsistent state shapes")
        lp = np.empty(B)
        comp = np.empty((B, 3)) if want_components else None
        _capi.check(L.mcd_prior_logprior_batch(self._p, *[_ptr(a) for a in arr], nn, B, 0, None, _ptr(lp),
                                               _ptr(comp) if want_components else None))
        return (lp, comp) if want_components else lp


    def grad(self, s: StateBatch):
        """ln prior and its gradient with respect to the seven fields of the state (host arrays):
        (lp [B], dict(time_birth_rate, time_death_rate, time_height, heights [B, n_nodes], rate_mean, rate_variance,
        rates [B, n_nodes])).  NaN outside the support."""
        if s.time_birth_rate is None or s.time_death_rate is None or s.rate_variance is None:
            raise ValueError("grad: the state batch lacks time_birth_rate / time_death_rate / rate_variance")
        nn = self.topo.n_nodes
        fields = (s.time_birth_rate, s.time_death_rate, s.time_height, s.heights, s.rate_mean, s.rate_variance, s.rates)
        arr = [np.ascontiguousarray(a, dtype=np.float64) for a in fields]
        B = arr[3].shape[0]
        if arr[3].shape != (B, nn) or arr[6].shape != (B, nn) or any(a.shape != (B,) for a in (arr[0], arr[1], arr[2], arr[4], arr[5])):
            raise ValueError("grad: inconsistent state shapes")
        lp, gb, gd, gt, gm, gv = (np.empty(B) for _ in range(6))
        gH, gR = np.empty((B, nn)), np.empty((B, nn))
        _capi.check(_capi.lib().mcd_prior_grad_batch(self._p, *[_ptr(a) for a in arr], nn, B, 0, None, _ptr(lp), _ptr(gb), _ptr(gd), _ptr(gt),
                                                     _ptr(gH), _ptr(gm), _ptr(gv), _ptr(gR)))
        return lp, dict(time_birth_rate=gb, time_death_rate=gd, time_height=gt, heights=gH, rate_mean=gm, rate_variance=gv, rates=gR)


def prior_function(ht: float, model: str, calibrations, constraints, braces, topo: Topology,
                   device: int = 0) -> Callable[[State], float]:
    """`priorFunction :: Double -> RelaxedMolecularClockModel -> ... -> PriorFunction I` (app/Probability.hs:127-150)."""
    pf = PriorFunction(ht, model, calibrations, constraints, braces, topo, device)

    def f(x: State) -> float:
        return float(pf.logprior(StateBatch.from_states([x]))[0])

    f.prior = pf
    return f
