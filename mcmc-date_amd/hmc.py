"""Device leapfrog of the Hamiltonian proposal (second part) and a plain HMC transition on top of it.

`Leapfrog` wraps `mcd_hmc_*` (include/mcmcdate_mvn.h): the state of B chains, their momenta and the gradient of
ln [prior x likelihood x jacobianRootBranch] (`htargetWith`, app/Hamiltonian.hs:72-92) live on the device; positions
use the reference's layout (`getMask` / `toVector`, :33-53; identical to `hamiltonian.to_vector`).

The reference's proposal is NUTS with step-size and mass tuning from the `mcmc` package (`nutsWith`, :95-105), which is
not restated here.  `hmc_transition` is the textbook fixed-length HMC step (momentum refresh, n leapfrog steps,
Metropolis correction) -- enough to exercise the integrator end to end and to check its stationary distribution
against Metropolis-Hastings chains with the same target.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from .likelihood import TreeLikelihood
from .prior import PriorFunction
from .state import StateBatch

_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


class Leapfrog:
    """B chains on one GPU.  `tree_lik` and `prior` must live on the same device and outlive this object."""

    def __init__(self, tree_lik: TreeLikelihood, prior: PriorFunction, calibrations_available: bool, batch: int):
        self.topo = tree_lik.topo
        self.batch = int(batch)
        self._keep = (tree_lik, prior)
        self._h = C.c_void_p()
        _capi.check(_capi.lib().mcd_hmc_create(C.byref(self._h), tree_lik._t, prior._p, int(bool(calibrations_available)), self.batch))
        self.dim = int(_capi.lib().mcd_hmc_dim(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _capi.lib().mcd_hmc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, s: StateBatch):
        nn = self.topo.n_nodes
        f = lambda x: np.ascontiguousarray(x, dtype=np.float64)
        arr = [f(s.time_birth_rate), f(s.time_death_rate), f(s.time_height), f(s.heights), f(s.rate_mean), f(s.rate_variance), f(s.rates)]
        if arr[3].shape != (self.batch, nn) or arr[6].shape != (self.batch, nn):
            raise ValueError("set_state: inconsistent state shapes")
        _capi.check(_capi.lib().mcd_hmc_set_state(self._h, *[_p(a) for a in arr], nn))

    def state(self) -> StateBatch:
        nn, B = self.topo.n_nodes, self.batch
        birth, death, tH, rMu, rVar = (np.empty(B) for _ in range(5))
        H, R = np.empty((B, nn)), np.empty((B, nn))
        _capi.check(_capi.lib().mcd_hmc_get_state(self._h, *[_p(a) for a in (birth, death, tH, H, rMu, rVar, R)], nn))
        return StateBatch(H, R, tH, rMu, birth, death, rVar)

    def position(self):
        """(q [B, dim], ln target [B], gradient [B, dim]) of the current state."""
        q, g = np.empty((self.batch, self.dim)), np.empty((self.batch, self.dim))
        v = np.empty(self.batch)
        _capi.check(_capi.lib().mcd_hmc_get_position(self._h, _p(q), _p(v), _p(g)))
        return q, v, g

    def leapfrog(self, p, eps, inv_mass, n_steps: int, direction=None):
        """n_steps leapfrog steps from the current state with momenta p [B, dim]; returns the new momenta (the state is
        advanced on the device; read it with position() / state())."""
        p = np.array(p, dtype=np.float64, order="C")
        eps = np.ascontiguousarray(np.broadcast_to(np.asarray(eps, np.float64), (self.batch,)))
        inv_mass = np.ascontiguousarray(np.broadcast_to(np.asarray(inv_mass, np.float64), (self.dim,)))
        d = None if direction is None else np.ascontiguousarray(direction, dtype=np.float64)
        if p.shape != (self.batch, self.dim):
            raise ValueError("leapfrog: p must be [batch, dim]")
        _capi.check(_capi.lib().mcd_hmc_leapfrog(self._h, _p(p), _p(eps), _p(d) if d is not None else None, _p(inv_mass), int(n_steps)))
        return p


def hmc_transition(lf: Leapfrog, rng: np.random.Generator, eps, inv_mass, n_steps: int) -> np.ndarray:
    """One fixed-length HMC transition for every chain: p ~ N(0, M), n leapfrog steps, accept with probability
    min(1, exp(H_old - H_new)), H = -ln target + 1/2 p^T M^-1 p.  Rejected chains (and chains that left the support:
    NaN) are put back.  Returns the acceptance mask."""
    inv_mass = np.broadcast_to(np.asarray(inv_mass, np.float64), (lf.dim,))
    old = lf.state()
    _, v0, _ = lf.position()
    p0 = rng.normal(size=(lf.batch, lf.dim)) / np.sqrt(inv_mass)
    h0 = -v0 + 0.5 * np.sum(p0 * p0 * inv_mass, axis=1)
    p1 = lf.leapfrog(p0, eps, inv_mass, n_steps)
    _, v1, _ = lf.position()
    h1 = -v1 + 0.5 * np.sum(p1 * p1 * inv_mass, axis=1)
    with np.errstate(over="ignore", invalid="ignore"):
        accept = np.log(rng.uniform(size=lf.batch)) < (h0 - h1)
    accept &= np.isfinite(h1)
    if not accept.all():
        new = lf.state()
        keep = accept
        merged = StateBatch(np.where(keep[:, None], new.heights, old.heights), np.where(keep[:, None], new.rates, old.rates),
                            np.where(keep, new.time_height, old.time_height), np.where(keep, new.rate_mean, old.rate_mean),
                            np.where(keep, new.time_birth_rate, old.time_birth_rate), np.where(keep, new.time_death_rate, old.time_death_rate),
                            np.where(keep, new.rate_variance, old.rate_variance))
        lf.set_state(merged)
    return accept
