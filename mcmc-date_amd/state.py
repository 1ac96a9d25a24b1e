"""The sampler state, mirroring `IG a` / `I` (app/State.hs:70-91).

A single `State` holds the seven fields of the reference record; trees are stored as per-node arrays
in pre-order (the Foldable order of HeightTree / LengthTree, lib/Mcmc/Tree/Types.hs:91-95, 146-150).
`StateBatch` is the many-chain form the device path consumes: chain-major arrays, one row per chain,
either numpy (host) or torch CUDA tensors (device resident).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List

import numpy as np

from .tree import Topology, is_valid_height_tree


@dataclass
class State:
    time_birth_rate: float          # _timeBirthRate
    time_death_rate: float          # _timeDeathRate
    time_height: float              # _timeHeight (absolute height of the time tree)
    time_tree: np.ndarray           # _timeTree :: HeightTree -- relative node heights, [n_nodes]
    rate_mean: float                # _rateMean
    rate_variance: float            # _rateVariance
    rate_tree: np.ndarray           # _rateTree :: LengthTree -- relative branch rates, [n_nodes], [0] = stem

    def is_valid(self, topo: Topology) -> bool:
        """isValidState -- app/State.hs:108-118."""
        r = np.asarray(self.rate_tree)
        return bool(
            self.time_birth_rate > 0 and self.time_death_rate > 0 and self.time_height > 0
            and is_valid_height_tree(topo, self.time_tree) and self.rate_mean > 0 and self.rate_variance > 0
            and r[0] >= 0 and np.all(r[1:] > 0)
        )


@dataclass
class StateBatch:
    """B chains; only the fields the likelihood reads (tH, heights, rMu, rates) are carried."""
    heights: "np.ndarray"        # [B, n_nodes]
    rates: "np.ndarray"          # [B, n_nodes]
    time_height: "np.ndarray"    # [B]
    rate_mean: "np.ndarray"      # [B]
    # the three fields only the prior reads (None when only the likelihood is evaluated)
    time_birth_rate: "np.ndarray" = None   # [B]
    time_death_rate: "np.ndarray" = None   # [B]
    rate_variance: "np.ndarray" = None     # [B]

    @classmethod
    def from_states(cls, xs: Iterable[State]) -> "StateBatch":
        xs = list(xs)
        return cls(
            np.stack([np.asarray(x.time_tree, np.float64) for x in xs]),
            np.stack([np.asarray(x.rate_tree, np.float64) for x in xs]),
            np.asarray([x.time_height for x in xs], np.float64),
            np.asarray([x.rate_mean for x in xs], np.float64),
            np.asarray([x.time_birth_rate for x in xs], np.float64),
            np.asarray([x.time_death_rate for x in xs], np.float64),
            np.asarray([x.rate_variance for x in xs], np.float64),
        )

    def __len__(self):
        return int(self.heights.shape[0])

    def to(self, device) -> "StateBatch":
        """Move to a torch device (float64, contiguous)."""
        import torch

        f = lambda a: torch.as_tensor(np.asarray(a) if not hasattr(a, "device") else a, dtype=torch.float64).to(device).contiguous()
        o = lambda a: None if a is None else f(a)
        return StateBatch(f(self.heights), f(self.rates), f(self.time_height), f(self.rate_mean), o(self.time_birth_rate),
                          o(self.time_death_rate), o(self.rate_variance))

    def slice(self, lo: int, hi: int) -> "StateBatch":
        o = lambda a: None if a is None else a[lo:hi]
        return StateBatch(self.heights[lo:hi], self.rates[lo:hi], self.time_height[lo:hi], self.rate_mean[lo:hi],
                          o(self.time_birth_rate), o(self.time_death_rate), o(self.rate_variance))
