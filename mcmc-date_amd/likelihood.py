"""Host-side mirror of the reference's likelihood plugin surface (app/Probability.hs:152-281,
361-410) on top of the C ABI (include/mcmcdate_mvn.h).

Names follow the reference: `LikelihoodData` with constructors `Full | Sparse | Univariate |
NoData` (app/Probability.hs:210-235), `likelihood_function` (:277-281), `jacobian_root_branch`
(:408-410), `.data` files (app/Main.hs:75-99).  All four constructors run through the same HIP
kernels: `Univariate` is a diagonal covariance, `Sparse` a precision matrix given as an association
list.  There is no CPU evaluation path here.

Arrays may be numpy (host pointers: copied, evaluated, returned synchronously) or torch CUDA
tensors (device pointers: enqueued on torch's current stream, results returned as CUDA tensors).
"""
from __future__ import annotations

import ctypes as C
import json
from dataclasses import dataclass
from typing import Callable, List, Sequence, Tuple, Union

import numpy as np

from . import _capi
from .state import State, StateBatch
from .tree import Topology

_dp = C.POINTER(C.c_double)


# ----------------------------------------------------------------------------------------------
# LikelihoodData -- app/Probability.hs:210-235
# ----------------------------------------------------------------------------------------------
@dataclass
class Full:
    """Multivariate normal with full inverted covariance matrix and log det of the covariance."""
    mu: np.ndarray
    sigma_inv: np.ndarray
    logdet_sigma: float


@dataclass
class Sparse:
    """Multivariate normal with sparse inverted covariance matrix, [((i, j), value), ...]."""
    mu: np.ndarray
    sigma_inv_assoc: List[Tuple[Tuple[int, int], float]]
    logdet_sigma: float


@dataclass
class Univariate:
    """Univariate normal distributions: means and variances."""
    mu: np.ndarray
    vs: np.ndarray


@dataclass
class NoData:
    """No likelihood; use prior only (likelihood 1.0, app/Probability.hs:281)."""


LikelihoodData = Union[Full, Sparse, Univariate, NoData]


def read_data_file(path: str) -> LikelihoodData:
    """`getData` -- app/Main.hs:85-99 (aeson default sum encoding of LikelihoodDataStore, :75-81)."""
    with open(path) as f:
        r = json.load(f)
    tag = r.get("tag")
    if tag == "FullS":
        mu, rows, logdet = r["contents"]
        return Full(np.asarray(mu, float), np.asarray(rows, float), float(logdet))
    if tag == "SparseS":
        mu, assoc, logdet = r["contents"]
        return Sparse(np.asarray(mu, float), [((int(ij[0]), int(ij[1])), float(v)) for ij, v in assoc], float(logdet))
    if tag == "UnivariateS":
        mu, vs = r["contents"]
        return Univariate(np.asarray(mu, float), np.asarray(vs, float))
    if tag == "NoLikelihoodS":
        return NoData()
    raise ValueError(f"getData: Could not decode data file: {path}.")  # app/Main.hs:89


def write_data_file(path: str, lhd: LikelihoodData) -> None:
    """Inverse of `read_data_file` (what `prepare` writes, app/Main.hs:240, 286)."""
    if isinstance(lhd, Full):
        obj = {"tag": "FullS", "contents": [list(map(float, lhd.mu)), [list(map(float, r)) for r in lhd.sigma_inv],
                                            float(lhd.logdet_sigma)]}
    elif isinstance(lhd, Sparse):
        obj = {"tag": "SparseS", "contents": [list(map(float, lhd.mu)),
                                              [[[int(i), int(j)], float(v)] for (i, j), v in lhd.sigma_inv_assoc],
                                              float(lhd.logdet_sigma)]}
    elif isinstance(lhd, Univariate):
        obj = {"tag": "UnivariateS", "contents": [list(map(float, lhd.mu)), list(map(float, lhd.vs))]}
    else:
        obj = {"tag": "NoLikelihoodS"}
    with open(path, "w") as f:
        json.dump(obj, f)


# ----------------------------------------------------------------------------------------------
# array plumbing
# ----------------------------------------------------------------------------------------------
def _is_torch(a) -> bool:
    return type(a).__module__.startswith("torch")


def _host(a, shape_tail=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a


def _ptr(a):
    if _is_torch(a):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(a.ctypes.data)


def _stream_ptr(dev_index: int):
    import torch

    return C.c_void_p(torch.cuda.current_stream(dev_index).cuda_stream)


def _check_cuda(t, device: int, name: str):
    import torch

    if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
        raise TypeError(f"{name}: need a contiguous float64 CUDA tensor")
    if t.device.index != device:
        raise ValueError(f"{name}: tensor is on cuda:{t.device.index}, likelihood lives on cuda:{device}")


# ----------------------------------------------------------------------------------------------
# the likelihood object (owns the C handle)
# ----------------------------------------------------------------------------------------------
FORMS = {"auto": 0, "sweep": 1, "multiply": 2}    # MCD_FORM_* (include/mcmcdate_mvn.h)


def set_logpdf_form(form: str) -> str:
    """Choose the form of the log-density kernels for this process: "auto" (by dimension and batch size), "sweep" (one
    chain per wave, the latency form) or "multiply" (z = L^-1 (x - mu) on the fp64 matrix cores, the throughput form).
    Returns the previous setting.  The two forms agree to rounding, not bit for bit."""
    if form not in FORMS:
        raise ValueError(f"set_logpdf_form: expected one of {sorted(FORMS)}, got {form!r}")
    prev = _capi.lib().mcd_set_logpdf_form(FORMS[form])
    if prev < 0:
        _capi.check(prev)
    return {v: k for k, v in FORMS.items()}[prev]


class MvnLikelihood:
    """`likelihoodFunction lhd` with its operands staged once on one GPU.

    Equivalent of the closure built in getLikelihoodFunction (app/Main.hs:333-347).
    """

    def __init__(self, lhd: LikelihoodData, device: int = 0):
        self._h = C.c_void_p()
        self.device = int(device)
        self.lhd = lhd
        self._nodata = isinstance(lhd, NoData)
        if self._nodata:
            self.n = 0
            return
        if isinstance(lhd, Full):
            mu = _host(lhd.mu)
            mat = _host(lhd.sigma_inv)
            kind, logdet = _capi.MCD_MAT_SIGMA_INV, float(lhd.logdet_sigma)
        elif isinstance(lhd, Sparse):
            mu = _host(lhd.mu)
            mat = np.zeros((len(mu), len(mu)))
            for (i, j), v in lhd.sigma_inv_assoc:      # L.mkSparse, app/Main.hs:95
                mat[i, j] += v
            kind, logdet = _capi.MCD_MAT_SIGMA_INV, float(lhd.logdet_sigma)
        elif isinstance(lhd, Univariate):
            mu = _host(lhd.mu)
            vs = _host(lhd.vs)
            mat = np.diag(vs)                           # Sigma = diag(vs); logdet = sum log vs (:274)
            kind, logdet = _capi.MCD_MAT_SIGMA, 0.0
        else:
            raise TypeError(f"not a LikelihoodData: {lhd!r}")
        n = len(mu)
        if mat.shape != (n, n):
            raise ValueError("LikelihoodData: matrix shape does not match the mean vector")
        self.n = n
        L = _capi.lib()
        _capi.check(L.mcd_mvn_create(C.byref(self._h), n, mu.ctypes.data_as(_dp), mat.ctypes.data_as(_dp), kind,
                                     C.c_double(logdet), self.device))

    @classmethod
    def from_covariance(cls, mu, sigma, device: int = 0) -> "MvnLikelihood":
        """Operands given as (mu, Sigma) -- what `meanCov` returns (app/Main.hs:208)."""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self.device = int(device)
        self.lhd = None
        self._nodata = False
        mu = _host(mu)
        sigma = _host(sigma)
        self.n = len(mu)
        if sigma.shape != (self.n, self.n):
            raise ValueError("from_covariance: matrix shape does not match the mean vector")
        _capi.check(_capi.lib().mcd_mvn_create(C.byref(self._h), self.n, mu.ctypes.data_as(_dp),
                                               sigma.ctypes.data_as(_dp), _capi.MCD_MAT_SIGMA, C.c_double(0.0),
                                               self.device))
        return self

    # -- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _capi.lib().mcd_mvn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def release_stream(self, stream) -> None:
        """mcd_mvn_release_stream: the scratch set the row-split kernels keep for `stream` (a torch.cuda.Stream or a raw
        hipStream_t value) goes back to the handle's pool; call it before destroying a short-lived stream."""
        raw = getattr(stream, "cuda_stream", stream)
        _capi.check(_capi.lib().mcd_mvn_release_stream(self._h, C.c_void_p(int(raw) if raw else 0)))

    def set_form(self, form: str) -> str:
        """Pin the form of the log-density kernels for THIS handle ("auto": follow the process default, "sweep",
        "multiply"); mcd_mvn_set_form.  Returns the previous choice."""
        if form not in FORMS:
            raise ValueError(f"set_form: expected one of {sorted(FORMS)}, got {form!r}")
        if self._nodata:
            return "auto"
        prev = _capi.lib().mcd_mvn_set_form(self._h, FORMS[form])
        if prev < 0:
            _capi.check(prev)
        return {v: k for k, v in FORMS.items()}[prev]

    @property
    def logdet_sigma(self) -> float:
        return 0.0 if self._nodata else float(_capi.lib().mcd_mvn_logdet(self._h))

    def cholesky_factor(self) -> np.ndarray:
        out = np.empty((self.n, self.n))
        _capi.check(_capi.lib().mcd_mvn_get_factor(self._h, out.ctypes.data_as(_dp)))
        return out

    # -- evaluation ----------------------------------------------------------------------------
    def logpdf1(self, x: Sequence[float]) -> float:
        """One evaluation: drop-in for logDensityFullMultivariateNormal (app/Probability.hs:166-173)."""
        if self._nodata:
            return 0.0
        x = _host(x)
        if x.shape != (self.n,):
            raise ValueError("logpdf1: wrong vector length")
        out = C.c_double()
        _capi.check(_capi.lib().mcd_mvn_logpdf(self._h, x.ctypes.data_as(_dp), C.byref(out)))
        return out.value

    def logpdf(self, X):
        """Batch of independent evaluations; X is [batch, n] chain-major (numpy or CUDA tensor)."""
        L = _capi.lib()
        if _is_torch(X):
            import torch

            if self._nodata:
                return torch.zeros(X.shape[0], dtype=torch.float64, device=X.device)
            _check_cuda(X, self.device, "X")
            if X.dim() != 2 or X.shape[1] != self.n:
                raise ValueError("logpdf: X must be [batch, n]")
            ll = torch.empty(X.shape[0], dtype=torch.float64, device=X.device)
            _capi.check(L.mcd_mvn_logpdf_batch(self._h, _ptr(X), X.stride(0), X.shape[0], 1, _stream_ptr(self.device),
                                               _ptr(ll)))
            return ll
        X = _host(X)
        if self._nodata:
            return np.zeros(X.shape[0])
        if X.ndim != 2 or X.shape[1] != self.n:
            raise ValueError("logpdf: X must be [batch, n]")
        ll = np.empty(X.shape[0])
        _capi.check(L.mcd_mvn_logpdf_batch(self._h, _ptr(X), X.shape[1], X.shape[0], 0, None, _ptr(ll)))
        return ll

    def logpdf_into(self, X, ll):
        """Device-resident variant writing into a caller-provided CUDA tensor (no allocation)."""
        _capi.check(_capi.lib().mcd_mvn_logpdf_batch(self._h, _ptr(X), X.stride(0), X.shape[0], 1,
                                                     _stream_ptr(self.device), _ptr(ll)))
        return ll

    def grad(self, X):
        """(ll, G) with G[b] = d ll / d x_b = -Sigma^-1 (x_b - mu)."""
        L = _capi.lib()
        if self._nodata:
            raise ValueError("grad: NoData has no gradient path (the reference offers likelihoodFunctionG for Full only)")
        if _is_torch(X):
            import torch

            _check_cuda(X, self.device, "X")
            ll = torch.empty(X.shape[0], dtype=torch.float64, device=X.device)
            G = torch.empty_like(X)
            _capi.check(L.mcd_mvn_grad_batch(self._h, _ptr(X), X.stride(0), X.shape[0], 1, _stream_ptr(self.device),
                                             _ptr(ll), _ptr(G), G.stride(0)))
            return ll, G
        X = _host(X)
        if X.ndim != 2 or X.shape[1] != self.n:
            raise ValueError("grad: X must be [batch, n]")
        ll = np.empty(X.shape[0])
        G = np.empty_like(X)
        _capi.check(L.mcd_mvn_grad_batch(self._h, _ptr(X), X.shape[1], X.shape[0], 0, None, _ptr(ll), _ptr(G), X.shape[1]))
        return ll, G

    def bind_tree(self, topo: Topology) -> "TreeLikelihood":
        return TreeLikelihood(self, topo)


class TreeLikelihood:
    """State -> log-likelihood: likelihoodFunctionWrapper (app/Probability.hs:195-207) on device."""

    def __init__(self, mvn: MvnLikelihood, topo: Topology):
        self.mvn = mvn
        self.topo = topo
        self._t = C.c_void_p()
        if mvn._nodata:
            topo.root_children()
            return
        par = np.ascontiguousarray(topo.parent, dtype=np.int32)
        _capi.check(_capi.lib().mcd_tree_create(C.byref(self._t), mvn._h, len(par),
                                                par.ctypes.data_as(C.POINTER(C.c_int32))))

    def close(self):
        if getattr(self, "_t", None) is not None and self._t.value:
            _capi.lib().mcd_tree_destroy(self._t)
            self._t = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _unpack(self, s: StateBatch):
        return s.heights, s.rates, s.time_height, s.rate_mean

    def loglik(self, states: StateBatch, want_jacobian: bool = True):
        """(ll[batch], log jacobianRootBranch[batch] or None)."""
        H, R, tH, rMu = self._unpack(states)
        L = _capi.lib()
        nn = self.topo.n_nodes
        if _is_torch(H):
            import torch

            dev = self.mvn.device
            for t, nm in ((H, "heights"), (R, "rates"), (tH, "time_height"), (rMu, "rate_mean")):
                _check_cuda(t, dev, nm)
            B = H.shape[0]
            if self.mvn._nodata:
                return torch.zeros(B, dtype=torch.float64, device=H.device), None
            ll = torch.empty(B, dtype=torch.float64, device=H.device)
            lj = torch.empty(B, dtype=torch.float64, device=H.device) if want_jacobian else None
            _capi.check(L.mcd_tree_loglik_batch(self._t, _ptr(H), _ptr(R), H.stride(0), _ptr(tH), _ptr(rMu), B, 1,
                                                _stream_ptr(dev), _ptr(ll), _ptr(lj) if want_jacobian else None))
            return ll, lj
        H, R, tH, rMu = (_host(a) for a in (H, R, tH, rMu))
        B = H.shape[0]
        if H.shape != (B, nn) or R.shape != (B, nn) or tH.shape != (B,) or rMu.shape != (B,):
            raise ValueError("loglik: inconsistent state shapes")
        if self.mvn._nodata:
            return np.zeros(B), None
        ll = np.empty(B)
        lj = np.empty(B) if want_jacobian else None
        _capi.check(L.mcd_tree_loglik_batch(self._t, _ptr(H), _ptr(R), nn, _ptr(tH), _ptr(rMu), B, 0, None, _ptr(ll),
                                            _ptr(lj) if want_jacobian else None))
        return ll, lj

    def grad(self, states: StateBatch):
        """(ll, g_heights[batch, n_nodes], g_rates[batch, n_nodes], g_time_height[batch], g_rate_mean[batch])."""
        if self.mvn._nodata:
            raise ValueError("grad: NoData has no gradient path")
        H, R, tH, rMu = self._unpack(states)
        L = _capi.lib()
        nn = self.topo.n_nodes
        if _is_torch(H):
            import torch

            dev = self.mvn.device
            for t, nm in ((H, "heights"), (R, "rates"), (tH, "time_height"), (rMu, "rate_mean")):
                _check_cuda(t, dev, nm)
            B = H.shape[0]
            ll = torch.empty(B, dtype=torch.float64, device=H.device)
            gH = torch.empty_like(H)
            gR = torch.empty_like(R)
            gt = torch.empty_like(tH)
            gm = torch.empty_like(rMu)
            _capi.check(L.mcd_tree_grad_batch(self._t, _ptr(H), _ptr(R), H.stride(0), _ptr(tH), _ptr(rMu), B, 1,
                                              _stream_ptr(dev), _ptr(ll), _ptr(gH), _ptr(gR), _ptr(gt), _ptr(gm)))
            return ll, gH, gR, gt, gm
        H, R, tH, rMu = (_host(a) for a in (H, R, tH, rMu))
        B = H.shape[0]
        if H.shape != (B, nn) or R.shape != (B, nn) or tH.shape != (B,) or rMu.shape != (B,):
            raise ValueError("grad: inconsistent state shapes")
        ll = np.empty(B)
        gH = np.empty_like(H)
        gR = np.empty_like(R)
        gt = np.empty(B)
        gm = np.empty(B)
        _capi.check(L.mcd_tree_grad_batch(self._t, _ptr(H), _ptr(R), nn, _ptr(tH), _ptr(rMu), B, 0, None, _ptr(ll),
                                          _ptr(gH), _ptr(gR), _ptr(gt), _ptr(gm)))
        return ll, gH, gR, gt, gm


# ----------------------------------------------------------------------------------------------
# the sparse form, natively (no densification; N up to MCD_MAX_SPARSE_DIM = 8192)
# ----------------------------------------------------------------------------------------------
MAX_SPARSE_DIM = 8192


class SparseLikelihood:
    """`likelihoodFunction (Sparse mu sigmaInvSparse logDetSigma)` (app/Probability.hs:279, 178-184) with the precision matrix kept
    sparse on the device (csrc/k_sparse.hip, mcd_sparse_*): the reference's route for trees with thousands of branches.  Takes the
    `Sparse` record of a `.data` file (read_data_file) or of `prepare`.  logpdf / grad on raw vectors, bind_tree(topo).loglik on
    states.  `MvnLikelihood(Sparse ...)` densifies instead (N <= 1024) and is what the samplers and the tree gradient use."""

    def __init__(self, lhd: Sparse, device: int = 0):
        if not isinstance(lhd, Sparse):
            raise TypeError("SparseLikelihood: need a Sparse record")
        self._h = C.c_void_p()
        self.device = int(device)
        self.lhd = lhd
        mu = _host(lhd.mu)
        self.n = len(mu)
        row = np.ascontiguousarray([ij[0] for ij, _ in lhd.sigma_inv_assoc], dtype=np.int32)
        col = np.ascontiguousarray([ij[1] for ij, _ in lhd.sigma_inv_assoc], dtype=np.int32)
        val = np.ascontiguousarray([v for _, v in lhd.sigma_inv_assoc], dtype=np.float64)
        ip = C.POINTER(C.c_int32)
        _capi.check(_capi.lib().mcd_sparse_create(C.byref(self._h), self.n, mu.ctypes.data_as(_dp), len(val), row.ctypes.data_as(ip),
                                                  col.ctypes.data_as(ip), val.ctypes.data_as(_dp), C.c_double(float(lhd.logdet_sigma)),
                                                  self.device))
        self.nnz = int(_capi.lib().mcd_sparse_nnz(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _capi.lib().mcd_sparse_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def logpdf(self, X):
        """[batch, n] chain-major (numpy or CUDA tensor) -> ll[batch]."""
        L = _capi.lib()
        if _is_torch(X):
            import torch

            _check_cuda(X, self.device, "X")
            if X.dim() != 2 or X.shape[1] != self.n:
                raise ValueError("logpdf: X must be [batch, n]")
            ll = torch.empty(X.shape[0], dtype=torch.float64, device=X.device)
            _capi.check(L.mcd_sparse_logpdf_batch(self._h, _ptr(X), X.stride(0), X.shape[0], 1, _stream_ptr(self.device), _ptr(ll)))
            return ll
        X = _host(X)
        if X.ndim != 2 or X.shape[1] != self.n:
            raise ValueError("logpdf: X must be [batch, n]")
        ll = np.empty(X.shape[0])
        _capi.check(L.mcd_sparse_logpdf_batch(self._h, _ptr(X), X.shape[1], X.shape[0], 0, None, _ptr(ll)))
        return ll

    def grad(self, X):
        """(ll, G) with G[b] = d ll / d x_b = -P (x_b - mu)."""
        L = _capi.lib()
        if _is_torch(X):
            import torch

            _check_cuda(X, self.device, "X")
            ll = torch.empty(X.shape[0], dtype=torch.float64, device=X.device)
            G = torch.empty_like(X)
            _capi.check(L.mcd_sparse_grad_batch(self._h, _ptr(X), X.stride(0), X.shape[0], 1, _stream_ptr(self.device), _ptr(ll), _ptr(G),
                                                G.stride(0)))
            return ll, G
        X = _host(X)
        if X.ndim != 2 or X.shape[1] != self.n:
            raise ValueError("grad: X must be [batch, n]")
        ll = np.empty(X.shape[0])
        G = np.empty_like(X)
        _capi.check(L.mcd_sparse_grad_batch(self._h, _ptr(X), X.shape[1], X.shape[0], 0, None, _ptr(ll), _ptr(G), X.shape[1]))
        return ll, G

    def bind_tree(self, topo: Topology) -> "SparseTreeLikelihood":
        return SparseTreeLikelihood(self, topo)


class SparseTreeLikelihood:
    """State -> ln likelihood and ln jacobianRootBranch over a sparse precision matrix (mcd_sparse_tree_*)."""

    def __init__(self, sp: SparseLikelihood, topo: Topology):
        self.sp = sp
        self.topo = topo
        self._t = C.c_void_p()
        par = np.ascontiguousarray(topo.parent, dtype=np.int32)
        _capi.check(_capi.lib().mcd_sparse_tree_create(C.byref(self._t), sp._h, len(par), par.ctypes.data_as(C.POINTER(C.c_int32))))

    def close(self):
        if getattr(self, "_t", None) is not None and self._t.value:
            _capi.lib().mcd_sparse_tree_destroy(self._t)
            self._t = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def loglik(self, states: StateBatch, want_jacobian: bool = True):
        """(ll[batch], log jacobianRootBranch[batch] or None)."""
        H, R, tH, rMu = states.heights, states.rates, states.time_height, states.rate_mean
        L = _capi.lib()
        nn = self.topo.n_nodes
        if _is_torch(H):
            import torch

            dev = self.sp.device
            for t, nm in ((H, "heights"), (R, "rates"), (tH, "time_height"), (rMu, "rate_mean")):
                _check_cuda(t, dev, nm)
            B = H.shape[0]
            ll = torch.empty(B, dtype=torch.float64, device=H.device)
            lj = torch.empty(B, dtype=torch.float64, device=H.device) if want_jacobian else None
            _capi.check(L.mcd_sparse_tree_loglik_batch(self._t, _ptr(H), _ptr(R), H.stride(0), _ptr(tH), _ptr(rMu), B, 1, _stream_ptr(dev),
                                                       _ptr(ll), _ptr(lj) if want_jacobian else None))
            return ll, lj
        H, R, tH, rMu = (_host(a) for a in (H, R, tH, rMu))
        B = H.shape[0]
        if H.shape != (B, nn) or R.shape != (B, nn) or tH.shape != (B,) or rMu.shape != (B,):
            raise ValueError("loglik: inconsistent state shapes")
        ll = np.empty(B)
        lj = np.empty(B) if want_jacobian else None
        _capi.check(L.mcd_sparse_tree_loglik_batch(self._t, _ptr(H), _ptr(R), nn, _ptr(tH), _ptr(rMu), B, 0, None, _ptr(ll),
                                                   _ptr(lj) if want_jacobian else None))
        return ll, lj


# ----------------------------------------------------------------------------------------------
# the reference's plugin functions
# ----------------------------------------------------------------------------------------------
def likelihood_function(lhd: LikelihoodData, topo: Topology, device: int = 0) -> Callable[[State], float]:
    """`likelihoodFunction :: LikelihoodData -> LikelihoodFunction I` (app/Probability.hs:277-281).

    Returns a pure function State -> log-likelihood (log domain) that evaluates on the GPU.
    """
    if isinstance(lhd, Sparse) and len(lhd.mu) > 1024:           # beyond the dense kernels: the precision matrix stays sparse on the device
        tl = SparseLikelihood(lhd, device).bind_tree(topo)
    else:
        try:
            tl = MvnLikelihood(lhd, device).bind_tree(topo)
        except _capi.NotPositiveDefinite:
            # The dense kernels need a factor of the matrix; the reference evaluates dx . (P dx) with whatever P the record holds
            # (app/Probability.hs:169, 183).  An indefinite precision matrix therefore takes the product form on the device
            # (csrc/k_sparse.hip: P as given, nothing factored) -- same value as the reference's, on the GPU.
            if isinstance(lhd, Full):
                P = _host(lhd.sigma_inv)
                ii, jj = np.nonzero(P)
                lhd = Sparse(lhd.mu, [((int(i), int(j)), float(P[i, j])) for i, j in zip(ii, jj)], lhd.logdet_sigma)
            elif not isinstance(lhd, Sparse):
                raise
            tl = SparseLikelihood(lhd, device).bind_tree(topo)

    def f(x: State) -> float:
        ll, _ = tl.loglik(StateBatch.from_states([x]), want_jacobian=False)
        return float(ll[0])

    f.tree_likelihood = tl
    return f


def jacobian_root_branch(lhd: LikelihoodData, topo: Topology, device: int = 0) -> Callable[[State], float]:
    """`jacobianRootBranch` (app/Probability.hs:408-410): log (1 / rootBranch x), evaluated on the GPU
    as a by-product of the likelihood kernel."""
    tl = MvnLikelihood(lhd, device).bind_tree(topo)

    def j(x: State) -> float:
        _, lj = tl.loglik(StateBatch.from_states([x]), want_jacobian=True)
        return float(lj[0])

    j.tree_likelihood = tl
    return j
