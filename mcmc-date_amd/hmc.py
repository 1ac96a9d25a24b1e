"""Device leapfrog of the Hamiltonian proposal (second part) and a plain HMC transition on top of it.

`Leapfrog` wraps `mcd_hmc_*` (include/mcmcdate_mvn.h): the state of B chains, their momenta and the gradient of
ln [prior x likelihood x jacobianRootBranch] (`htargetWith`, app/Hamiltonian.hs:72-92) live on the device; positions
use the reference's layout (`getMask` / `toVector`, :33-53; identical to `hamiltonian.to_vector`).

The reference's proposal is NUTS from the `mcmc` package (`nutsWith`, :95-105; dschrempf/mcmc 542c43f6, not vendored).
Restated here from the published algorithm it implements -- Hoffman & Gelman, "The No-U-Turn Sampler", JMLR 15 (2014):
`nuts_transition` is Algorithm 3 (efficient NUTS: slice variable, recursive doubling, U-turn and divergence stops,
uniform sampling from the admissible set), `DualAveraging` the step-size adaptation of Algorithm 6 -- with the device
doing every leapfrog step for all chains at once and the per-chain recursion as host-side control flow (one Python
generator per chain; the driver gathers the leapfrog requests of all active chains into one `mcd_hmc_step_from` call).
The product path is `Leapfrog.nuts` / `Leapfrog.nuts_run`: the same algorithm as a per-chain state machine ON THE DEVICE
(csrc/k_nuts.hip, mcd_hmc_nuts*), dual averaging inside the library, step-level parity with a CPU twin
(tests/test_gpu_nuts.py); the host-side recursion below stays as the readable restatement of the control flow.
`mcmc`'s own tuning (`HTuneLeapfrog HTuneAllMasses`) and its defaults are NOT restated: parity unpinned for this part.
`hmc_transition` is the textbook fixed-length HMC step, kept as the simplest end-to-end exercise of the integrator.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

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


    def nuts(self, eps, inv_mass, max_depth: int = 8, seed: int = 0, transition: int = 0, chain_offset: int = 0):
        """One NUTS transition of every chain ON THE DEVICE (mcd_hmc_nuts, csrc/k_nuts.hip: Hoffman & Gelman 2014, Algorithm 3
        as a per-chain state machine; counter-based random streams (seed, chain_offset + b, transition)).  Returns (mean
        acceptance statistic [B], tree depth [B]); the chains' new states are on the device."""
        eps = np.ascontiguousarray(np.broadcast_to(np.asarray(eps, np.float64), (self.batch,)))
        inv_mass = np.ascontiguousarray(np.broadcast_to(np.asarray(inv_mass, np.float64), (self.dim,)))
        alpha = np.empty(self.batch)
        depth = np.empty(self.batch, np.int32)
        _capi.check(_capi.lib().mcd_hmc_nuts(self._h, _p(eps), _p(inv_mass), int(max_depth), int(seed), int(chain_offset), int(transition),
                                             _p(alpha), depth.ctypes.data_as(C.POINTER(C.c_int32))))
        return alpha, depth

    def nuts_run(self, n_transitions: int, eps, inv_mass, adapt: bool = False, delta: float = 0.65, max_depth: int = 8, seed: int = 0,
                 first_transition: int = 0, chain_offset: int = 0):
        """n transitions in the library (mcd_hmc_nuts_run); adapt: dual averaging of the step sizes (Algorithm 6).  Returns
        (eps [B] -- the averaged step sizes when adapting --, mean acceptance statistic [B], position means [dim], pooled
        position variances [dim])."""
        eps = np.array(np.broadcast_to(np.asarray(eps, np.float64), (self.batch,)), dtype=np.float64, order="C")
        inv_mass = np.ascontiguousarray(np.broadcast_to(np.asarray(inv_mass, np.float64), (self.dim,)))
        ma, qm, qv = np.empty(self.batch), np.empty(self.dim), np.empty(self.dim)
        _capi.check(_capi.lib().mcd_hmc_nuts_run(self._h, int(n_transitions), int(bool(adapt)), _p(eps), _p(inv_mass), float(delta), int(max_depth),
                                                 int(seed), int(chain_offset), int(first_transition), _p(ma), _p(qm), _p(qv)))
        return eps, ma, qm, qv

    def nuts_warmup(self, eps, inv_mass, windows: int = 3, window: int = 60, delta: float = 0.65, max_depth: int = 6, seed: int = 0,
                    first_transition: int = 0, chain_offset: int = 0):
        """Step sizes and diagonal masses tuned in the library (mcd_hmc_nuts_warmup: `HTuneLeapfrog HTuneAllMasses`,
        app/Hamiltonian.hs:62-63).  Returns (eps [B], inv_mass [dim], mean acceptance statistic of the closing window [B])."""
        eps = np.array(np.broadcast_to(np.asarray(eps, np.float64), (self.batch,)), dtype=np.float64, order="C")
        inv_mass = np.array(np.broadcast_to(np.asarray(inv_mass, np.float64), (self.dim,)), dtype=np.float64, order="C")
        ma = np.empty(self.batch)
        _capi.check(_capi.lib().mcd_hmc_nuts_warmup(self._h, int(windows), int(window), _p(eps), _p(inv_mass), float(delta), int(max_depth), int(seed),
                                                    int(chain_offset), int(first_transition), _p(ma)))
        return eps, inv_mass, ma

    def step_from(self, q, p, grad, eps, inv_mass, direction=None, have_grad=True):
        """One leapfrog step from the given phase points (all [B, dim]); returns (q', p', grad', ln target')."""
        q = np.array(q, dtype=np.float64, order="C")
        p = np.array(p, dtype=np.float64, order="C")
        g = np.array(grad, dtype=np.float64, order="C")
        eps = np.ascontiguousarray(np.broadcast_to(np.asarray(eps, np.float64), (self.batch,)))
        inv_mass = np.ascontiguousarray(np.broadcast_to(np.asarray(inv_mass, np.float64), (self.dim,)))
        d = None if direction is None else np.ascontiguousarray(direction, dtype=np.float64)
        v = np.empty(self.batch)
        _capi.check(_capi.lib().mcd_hmc_step_from(self._h, _p(q), _p(p), _p(g), int(bool(have_grad)), _p(eps),
                                                  _p(d) if d is not None else None, _p(inv_mass), _p(v)))
        return q, p, g, v


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


# ---- NUTS: Hoffman & Gelman (2014), Algorithm 3, one generator per chain -----------------------------------------------
DELTA_MAX = 1000.0


def _nuts_chain(q0, g0, logp0, r0, log_u, rng, inv_mass, max_depth, stats):
    """Generator of one chain's transition.  Yields leapfrog requests (q, r, grad, direction) and receives
    (q', r', grad', ln target'); returns the selected (q, grad, ln target)."""

    def kinetic(r):
        return 0.5 * float(np.sum(r * r * inv_mass))

    def no_u_turn(qm, rm, qp, rp):
        d = qp - qm
        return float(np.dot(d, rm * inv_mass)) >= 0.0 and float(np.dot(d, rp * inv_mass)) >= 0.0

    def build_tree(q, r, g, v, j):
        if j == 0:
            q1, r1, g1, lp1 = yield (q, r, g, v)
            joint = lp1 - kinetic(r1)                       # -H
            if not math.isfinite(joint):
                joint = -math.inf
            n1 = 1 if log_u <= joint else 0
            s1 = log_u < DELTA_MAX + joint
            stats["alpha"] += min(1.0, math.exp(min(0.0, joint - stats["joint0"])))
            stats["n_alpha"] += 1
            return q1, r1, g1, q1, r1, g1, q1, g1, lp1, n1, s1
        qm, rm, gm, qp, rp, gp, qc, gc, lpc, n1, s1 = yield from build_tree(q, r, g, v, j - 1)
        if s1:
            if v == -1:
                qm, rm, gm, _, _, _, qc2, gc2, lpc2, n2, s2 = yield from build_tree(qm, rm, gm, v, j - 1)
            else:
                _, _, _, qp, rp, gp, qc2, gc2, lpc2, n2, s2 = yield from build_tree(qp, rp, gp, v, j - 1)
            if n2 > 0 and rng.uniform() < n2 / max(1, n1 + n2):
                qc, gc, lpc = qc2, gc2, lpc2
            s1 = s2 and no_u_turn(qm, rm, qp, rp)
            n1 += n2
        return qm, rm, gm, qp, rp, gp, qc, gc, lpc, n1, s1

    qm = qp = q0
    rm = rp = r0
    gm = gp = g0
    q_new, g_new, lp_new = q0, g0, logp0
    n, s, j = 1, True, 0
    while s and j < max_depth:
        v = -1 if rng.uniform() < 0.5 else 1
        if v == -1:
            qm, rm, gm, _, _, _, qc, gc, lpc, n1, s1 = yield from build_tree(qm, rm, gm, v, j)
        else:
            _, _, _, qp, rp, gp, qc, gc, lpc, n1, s1 = yield from build_tree(qp, rp, gp, v, j)
        if s1 and rng.uniform() < min(1.0, n1 / n):
            q_new, g_new, lp_new = qc, gc, lpc
        n += n1
        s = s1 and no_u_turn(qm, rm, qp, rp)
        j += 1
    stats["depth"] = j
    return q_new, g_new, lp_new


def nuts_transition(lf: Leapfrog, rng: np.random.Generator, eps, inv_mass, max_depth: int = 8):
    """One NUTS transition (Algorithm 3) for every chain of `lf`, all chains advancing their trees in lock step on the
    device.  eps: scalar or [B].  Returns (mean acceptance statistic alpha / n_alpha per chain [B], tree depth [B]);
    the chains' new states are on the device (lf.state(), lf.position())."""
    B, D = lf.batch, lf.dim
    inv_mass = np.ascontiguousarray(np.broadcast_to(np.asarray(inv_mass, np.float64), (D,)))
    eps = np.ascontiguousarray(np.broadcast_to(np.asarray(eps, np.float64), (B,)))
    q0, lp0, g0 = lf.position()
    r0 = rng.normal(size=(B, D)) / np.sqrt(inv_mass)
    joint0 = lp0 - 0.5 * np.sum(r0 * r0 * inv_mass, axis=1)
    log_u = joint0 + np.log(rng.uniform(size=B))
    chain_rngs = [np.random.default_rng(rng.integers(0, 2 ** 63)) for _ in range(B)]
    stats = [{"alpha": 0.0, "n_alpha": 0, "joint0": float(joint0[b]), "depth": 0} for b in range(B)]
    gens = [_nuts_chain(q0[b].copy(), g0[b].copy(), float(lp0[b]), r0[b].copy(), float(log_u[b]), chain_rngs[b], inv_mass, max_depth, stats[b])
            for b in range(B)]
    result = [None] * B
    request = [None] * B
    for b in range(B):
        try:
            request[b] = next(gens[b])
        except StopIteration as done:               # cannot happen before the first leapfrog, kept for completeness
            result[b] = done.value
    q_in, p_in, g_in = q0.copy(), r0.copy(), g0.copy()
    direction = np.ones(B)
    while any(r is not None for r in request):
        for b in range(B):
            if request[b] is not None:
                q_in[b], p_in[b], g_in[b], v = request[b]
                direction[b] = float(v)
            else:                                    # finished chains ride along with a harmless step from their result
                q_in[b], g_in[b] = result[b][0], result[b][1]
                p_in[b] = 0.0
                direction[b] = 1.0
        q1, p1, g1, lp1 = lf.step_from(q_in, p_in, g_in, eps, inv_mass, direction=direction, have_grad=True)
        for b in range(B):
            if request[b] is None:
                continue
            try:
                request[b] = gens[b].send((q1[b].copy(), p1[b].copy(), g1[b].copy(), float(lp1[b])))
            except StopIteration as done:
                request[b] = None
                result[b] = done.value
    # the selected points become the chains' states (zero-length step: the device evaluates ln target and gradient there)
    q_sel = np.array([r[0] for r in result])
    lf.step_from(q_sel, np.zeros((B, D)), np.array([r[1] for r in result]), np.zeros(B), inv_mass, have_grad=True)
    alpha = np.array([s["alpha"] / max(1, s["n_alpha"]) for s in stats])
    return alpha, np.array([s["depth"] for s in stats])


class DualAveraging:
    """Step-size adaptation of Hoffman & Gelman, Algorithm 6 (one instance per chain set; vectorised over chains)."""

    def __init__(self, eps0, delta: float = 0.65, gamma: float = 0.05, t0: float = 10.0, kappa: float = 0.75):
        eps0 = np.asarray(eps0, np.float64)
        self.mu = np.log(10.0 * eps0)
        self.delta, self.gamma, self.t0, self.kappa = delta, gamma, t0, kappa
        self.h_bar = np.zeros_like(eps0)
        self.log_eps = np.log(eps0)
        self.log_eps_bar = np.zeros_like(eps0)
        self.m = 0

    def update(self, alpha):
        self.m += 1
        m = self.m
        self.h_bar = (1.0 - 1.0 / (m + self.t0)) * self.h_bar + (self.delta - np.asarray(alpha)) / (m + self.t0)
        self.log_eps = self.mu - math.sqrt(m) / self.gamma * self.h_bar
        w = m ** (-self.kappa)
        self.log_eps_bar = w * self.log_eps + (1.0 - w) * self.log_eps_bar
        return np.exp(self.log_eps)

    def final(self):
        return np.exp(self.log_eps_bar)


def nuts_warmup(lf: Leapfrog, rng: np.random.Generator, n_windows: int = 3, window: int = 60, eps0: float = 0.02, delta: float = 0.65,
                max_depth: int = 6):
    """Warm-up of step sizes and diagonal masses (the reference tunes both: `HTuningConf HTuneLeapfrog HTuneAllMasses`,
    app/Hamiltonian.hs:62-63; `mcmc`'s own scheme is not restated).  Windows of `window` transitions: dual averaging
    of the step sizes inside a window, then the inverse masses are set to the pooled variance of the positions visited
    in that window (over chains and transitions, shrunk towards their mean for stability).  The first window starts
    from inverse masses (0.1 q)^2.  Returns (eps [B], inv_mass [dim])."""
    q0, _, _ = lf.position()
    inv_mass = np.maximum((0.1 * np.abs(q0)).mean(axis=0) ** 2, 1e-12)
    eps = np.full(lf.batch, float(eps0))
    for w in range(n_windows):
        da = DualAveraging(eps, delta=delta)
        qs = []
        for _ in range(window):
            alpha, _ = nuts_transition(lf, rng, eps, inv_mass, max_depth=max_depth)
            eps = da.update(alpha)
            qs.append(lf.position()[0])
        eps = da.final()
        var = np.concatenate(qs[window // 4:]).var(axis=0)
        n_eff = lf.batch * (window - window // 4)
        inv_mass = (n_eff / (n_eff + 5.0)) * var + 1e-3 * (5.0 / (n_eff + 5.0))      # Stan-style shrinkage
    da = DualAveraging(eps, delta=delta)
    for _ in range(window):
        alpha, _ = nuts_transition(lf, rng, eps, inv_mass, max_depth=max_depth)
        eps = da.update(alpha)
    return da.final(), inv_mass


NUTS_STREAM_DOMAIN = 0x4E5554535F524E47          # "NUTS_RNG": keeps the NUTS streams apart from the Metropolis-Hastings streams of the same seed


def run_cycle_with_nuts(sampler, lf: Leapfrog, rng: np.random.Generator, n_iter: int, eps, inv_mass, max_depth: int = 6,
                        accumulate: bool = False, device: bool = True, seed: Optional[int] = None, first_transition: Optional[int] = None,
                        chain_offset: Optional[int] = None):
    """The reference's `--hamiltonian` mode: the Metropolis-Hastings cycle plus one NUTS proposal per iteration
    (`maybeHamiltonianProposal`, weight 1, app/Definitions.hs:272-274, 104-105 of app/Hamiltonian.hs).  The cycle runs in the lock-step
    driver, the NUTS transition on the device (`device=True`: Leapfrog.nuts, tree building in csrc/k_nuts.hip; False: the
    host-side recursion nuts_transition, kept as the reference implementation of the control flow); the states move between
    the two handles through the host once per iteration (a few KB).  The reference shuffles the NUTS proposal into the
    cycle; here it closes every iteration.

    Random streams of the device transitions: Philox (seed ^ NUTS_STREAM_DOMAIN, chain_offset + b, transition).  `seed`
    defaults to the sampler's (the domain constant keeps the draws apart from the Metropolis-Hastings draws of the same
    (chain, step)); `chain_offset` to the sampler's first GLOBAL chain (shards.shard_sampler), so the ranks of a sharded run draw
    different numbers; `first_transition` to the number of transitions this Leapfrog has already made through this
    function, so a burn-in call followed by a sampling call does not replay its draws.
    Returns (mean acceptance statistic of the NUTS transitions, mean absolute node ages tH * h_v over chains and
    iterations [n_nodes] or None)."""
    seed = int(getattr(sampler, "seed", 0) if seed is None else seed) ^ NUTS_STREAM_DOMAIN
    chain_offset = int(getattr(sampler, "first_chain", 0) if chain_offset is None else chain_offset)
    t0 = int(getattr(lf, "cycle_transitions_done", 0) if first_transition is None else first_transition)
    alphas = []
    ages = np.zeros(lf.topo.n_nodes) if accumulate else None
    for it in range(n_iter):
        sampler.run(1)
        lf.set_state(sampler.state())
        if device:
            alpha, _ = lf.nuts(eps, inv_mass, max_depth=max_depth, seed=seed, transition=t0 + it, chain_offset=chain_offset)
        else:
            alpha, _ = nuts_transition(lf, rng, eps, inv_mass, max_depth=max_depth)
        alphas.append(alpha.mean())
        s = lf.state()
        sampler.set_state(s)
        if accumulate:
            ages += (s.time_height[:, None] * s.heights).mean(axis=0)
    lf.cycle_transitions_done = t0 + n_iter
    return (float(np.mean(alphas)) if alphas else float("nan")), (ages / max(1, n_iter) if accumulate else None)
