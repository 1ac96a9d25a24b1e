"""Host-side glue of the Hamiltonian proposal, mirroring app/Hamiltonian.hs:33-60.

The reference's NUTS proposal works on a flat position vector: the masked `toList` of the state record
`IG` in REVERSE fold order (`toVector` conses while folding).  Fold order of `IG` (app/State.hs:70-100):
timeBirthRate, timeDeathRate, timeHeight, timeTree heights (pre-order), rateMean, rateVariance,
rateTree branches (pre-order).  Masked out (`getMask`, :33-47): the root height of the time tree, all leaf
heights, the stem of the rate tree, and timeHeight when no calibrations are available.

`grad_to_vector` arranges the device gradient of the LIKELIHOOD (mcd_tree_grad_batch) in that position
vector; the likelihood does not depend on timeBirthRate, timeDeathRate, rateVariance, so their entries are 0
(their gradients come from the prior: `PriorFunction.grad`, mcd_prior_grad_batch).  `target_grad` assembles the gradient
of the whole Hamiltonian target ln (prior x likelihood x jacobianRootBranch) (`htargetWith`, :72-92) for a batch.
"""
from __future__ import annotations

import numpy as np

from .state import State
from .tree import Topology


def get_mask(calibrations_available: bool, topo: Topology) -> np.ndarray:
    """`getMask` -- app/Hamiltonian.hs:33-47, in fold order of `IG`."""
    nn = topo.n_nodes
    heights = ~topo.leaves                      # leaves: False
    heights[0] = False                          # root height of the relative time tree
    rates = np.ones(nn, bool)
    rates[0] = False                            # stem of the rate tree
    return np.concatenate([[True, True, bool(calibrations_available)], heights, [True, True], rates])


def _fold(x: State) -> np.ndarray:
    return np.concatenate([[x.time_birth_rate, x.time_death_rate, x.time_height], np.asarray(x.time_tree, float),
                           [x.rate_mean, x.rate_variance], np.asarray(x.rate_tree, float)])


def to_vector(mask: np.ndarray, x: State) -> np.ndarray:
    """`toVector` -- app/Hamiltonian.hs:49-53 (reverse fold order of the unmasked entries)."""
    v = _fold(x)
    if len(mask) != len(v):
        raise ValueError("toVector: Mask is too short.")
    return v[mask][::-1].copy()


def from_vector_with(mask: np.ndarray, x: State, xs: np.ndarray) -> State:
    """`fromVectorWith` -- app/Hamiltonian.hs:55-60 (refill from the last index down)."""
    v = _fold(x)
    if len(mask) != len(v) or int(mask.sum()) != len(xs):
        raise ValueError("fromVectorWith: Mask is too short or traversable structure is too long.")
    v[mask] = np.asarray(xs, float)[::-1]
    nn = len(x.time_tree)
    return State(v[0], v[1], v[2], v[3:3 + nn].copy(), v[3 + nn], v[4 + nn], v[5 + nn:].copy())


def grad_to_vector(mask: np.ndarray, g_heights, g_rates, g_time_height: float, g_rate_mean: float) -> np.ndarray:
    """Likelihood gradient (one chain) in the position-vector layout of `to_vector`."""
    full = np.concatenate([[0.0, 0.0, g_time_height], np.asarray(g_heights, float), [g_rate_mean, 0.0],
                           np.asarray(g_rates, float)])
    return full[mask][::-1].copy()


def target_grad(mask: np.ndarray, tree_lik, prior, states):
    """Value and gradient of the Hamiltonian target of the reference, ln [prior x likelihood x jacobianRootBranch]
    (`htargetWith`, app/Hamiltonian.hs:72-92), for every chain of `states`, in the position-vector layout of
    `to_vector`: (value [B], gradient [B, mask.sum()]).  Prior and likelihood (values and gradients) come from the device
    (mcd_prior_grad_batch, mcd_tree_grad_batch); the Jacobian factor 1 / rootBranch (app/Probability.hs:393-410),
    rootBranch = tH rMu (t_l r_l + t_r r_r), is five numbers per chain and is differentiated here."""
    lp, gp = prior.grad(states)
    ll, gH, gR, gt, gm = tree_lik.grad(states)
    lj = tree_lik.loglik(states)[1]
    H, R = np.asarray(states.heights), np.asarray(states.rates)
    tH, rMu = np.asarray(states.time_height), np.asarray(states.rate_mean)
    topo = tree_lik.topo
    l, r = topo.root_children()
    S = (H[:, 0] - H[:, l]) * R[:, l] + (H[:, 0] - H[:, r]) * R[:, r]
    jH = np.zeros_like(H)
    jR = np.zeros_like(R)
    jH[:, l] = R[:, l] / S                                    # d/d h_l of -ln S
    jH[:, r] = R[:, r] / S
    jH[:, 0] = -(R[:, l] + R[:, r]) / S
    jR[:, l] = -(H[:, 0] - H[:, l]) / S
    jR[:, r] = -(H[:, 0] - H[:, r]) / S
    B = H.shape[0]
    full = np.concatenate([gp["time_birth_rate"][:, None], gp["time_death_rate"][:, None], (gp["time_height"] + gt - 1.0 / tH)[:, None],
                           gp["heights"] + gH + jH, (gp["rate_mean"] + gm - 1.0 / rMu)[:, None], gp["rate_variance"][:, None],
                           gp["rates"] + gR + jR], axis=1)
    assert full.shape == (B, len(mask))
    return lp + ll + lj, full[:, mask][:, ::-1].copy()
