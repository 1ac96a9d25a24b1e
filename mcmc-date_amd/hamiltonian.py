"""Host-side glue of the Hamiltonian proposal, mirroring app/Hamiltonian.hs:33-60.

The reference's NUTS proposal works on a flat position vector: the masked `toList` of the state record
`IG` in REVERSE fold order (`toVector` conses while folding).  Fold order of `IG` (app/State.hs:70-100):
timeBirthRate, timeDeathRate, timeHeight, timeTree heights (pre-order), rateMean, rateVariance,
rateTree branches (pre-order).  Masked out (`getMask`, :33-47): the root height of the time tree, all leaf
heights, the stem of the rate tree, and timeHeight when no calibrations are available.

`grad_to_vector` arranges the device gradient of the LIKELIHOOD (mcd_tree_grad_batch) in that position
vector; the likelihood does not depend on timeBirthRate, timeDeathRate, rateVariance, so their entries are 0
(their gradients come from the prior, which is outside this repository's scope).
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
