"""Monitors and node-age summaries of the batched sampler (reporting part).

Host-side mirror of
  * the monitor files of app/Definitions.hs:288-417 -- `params` (the five scalars, calibrated node ages, constrained
    node age differences, brace variances), `timetree` (absolute time tree), `ratetree`, `prior` (the three prior
    blocks), all with period 2 -- written per chain in the tab-separated layout of `mcmc`'s file monitors
    (first column `Iteration`);
  * the node-age summary of scripts/trees-monitor-summary-ultrametric:149-175: after dropping round(l * burn-in)
    samples, per node the mean, the maximum-likelihood variance, minimum, maximum and the 95 % interval taken from
    the sorted ages as slice(floor(0.025 l), floor(0.95 l)).

The states come from the device sampler (`Sampler.state()` every `period` iterations); nothing here computes a
likelihood or a prior on the host -- the prior blocks are evaluated by the device prior (`PriorFunction.logprior`).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from .prepare import to_newick
from .state import StateBatch
from .tree import Topology, height_tree_to_length_tree

PERIOD = 2   # every monitor file of the reference uses period 2 (app/Definitions.hs:358, 369, 372, 389)


@dataclass
class Trace:
    """Sampled states of all chains: arrays [n_samples, B, ...]; `iteration[k]` is the iteration of sample k."""
    iteration: np.ndarray
    time_birth_rate: np.ndarray
    time_death_rate: np.ndarray
    time_height: np.ndarray
    heights: np.ndarray
    rate_mean: np.ndarray
    rate_variance: np.ndarray
    rates: np.ndarray

    def ages(self) -> np.ndarray:
        """Absolute node ages tH * h_v, [n_samples, B, n_nodes] (getTimeTreeNodeHeight, app/Definitions.hs:300-304)."""
        return self.time_height[:, :, None] * self.heights

    def states(self, k: int) -> StateBatch:
        return StateBatch(self.heights[k], self.rates[k], self.time_height[k], self.rate_mean[k], self.time_birth_rate[k],
                          self.time_death_rate[k], self.rate_variance[k])


def collect(sampler, n_iter: int, period: int = PERIOD, accumulate: bool = False) -> Trace:
    """Advance the sampler by n_iter iterations and keep the state of every chain every `period` iterations."""
    cols: List[List[np.ndarray]] = [[] for _ in range(7)]
    its = []
    done = 0
    while done + period <= n_iter:
        sampler.run(period, accumulate=accumulate)
        done += period
        s = sampler.state()
        for col, a in zip(cols, (s.time_birth_rate, s.time_death_rate, s.time_height, s.heights, s.rate_mean, s.rate_variance, s.rates)):
            col.append(a)
        its.append(sampler.iterations_done)
    if done < n_iter:
        sampler.run(n_iter - done, accumulate=accumulate)
    st = [np.stack(c) if c else np.empty((0,)) for c in cols]
    return Trace(np.asarray(its, np.int64), *st)


# ---- monitor files ---------------------------------------------------------------------------------------------------
def _fmt(x: float) -> str:
    return repr(float(x))


def write_monitor_files(prefix: str, trace: Trace, chain: int, topo: Topology, calibrations: Sequence = (), constraints: Sequence = (),
                        braces: Sequence = (), prior=None) -> List[str]:
    """Write <prefix>.params.monitor, .timetree.monitor, .ratetree.monitor (and .prior.monitor when the device prior
    `prior` is given) for one chain.  Returns the file names."""
    files = []
    ages = trace.ages()[:, chain, :]
    names = (["TimeBirthRate", "TimeDeathRate", "TimeHeight", "RateMean", "RateVariance"]
             + [f"Calibration {c.name} ({c.lower if c.lower is not None else 0.0},{c.upper if c.upper is not None else 'Infinity'})" for c in calibrations]
             + [f"Constraint {k.name}" for k in constraints] + [f"Brace {b.name} variance" for b in braces])
    fn = prefix + ".params.monitor"
    with open(fn, "w") as f:
        f.write("\t".join(["Iteration"] + names) + "\n")
        for k, it in enumerate(trace.iteration):
            row = [trace.time_birth_rate[k, chain], trace.time_death_rate[k, chain], trace.time_height[k, chain], trace.rate_mean[k, chain],
                   trace.rate_variance[k, chain]]
            row += [ages[k, c.node] for c in calibrations]
            row += [ages[k, c.old] - ages[k, c.young] for c in constraints]                 # getTimeTreeDeltaNodeHeight, :321-322
            row += [float(np.var(ages[k, list(b.nodes)], ddof=1)) for b in braces]           # S.variance (unbiased), :335-339
            f.write("\t".join([str(int(it))] + [_fmt(x) for x in row]) + "\n")
    files.append(fn)
    for tag, col in (("timetree", "TimeTree"), ("ratetree", "RateTree")):
        fn = f"{prefix}.{tag}.monitor"
        with open(fn, "w") as f:
            f.write(f"Iteration\t{col}\n")
            for k, it in enumerate(trace.iteration):
                if tag == "timetree":     # absoluteTimeTree: heightTreeToLengthTree scaled by the time height, :360-364
                    lengths = height_tree_to_length_tree(topo, trace.heights[k, chain]) * trace.time_height[k, chain]
                else:
                    lengths = trace.rates[k, chain]
                f.write(f"{int(it)}\t{to_newick(topo, lengths)}\n")
        files.append(fn)
    if prior is not None:
        fn = prefix + ".prior.monitor"
        with open(fn, "w") as f:
            f.write("Iteration\tPriorCsKsBs\tPriorBirthDeath\tPriorRelaxedMolecularClock\n")
            for k, it in enumerate(trace.iteration):
                _, comp = prior.logprior(trace.states(k).slice(chain, chain + 1), want_components=True)
                f.write("\t".join([str(int(it))] + [_fmt(x) for x in comp[0]]) + "\n")
        files.append(fn)
    return files


# ---- node-age summary -- scripts/trees-monitor-summary-ultrametric:149-175, 222-255 --------------------------------------
SUMMARY_HEADER = "Index\tName\tMean\tVariance\tMin\tMax\t95CILower\t95CIUpper"


@dataclass
class AgeSummary:
    index: np.ndarray
    name: List[str]
    mean: np.ndarray
    variance: np.ndarray
    minimum: np.ndarray
    maximum: np.ndarray
    ci_lower: np.ndarray
    ci_upper: np.ndarray

    def render(self) -> str:
        rows = [SUMMARY_HEADER]
        for i in range(len(self.index)):
            rows.append("\t".join([str(int(self.index[i])), self.name[i]] + [_fmt(x) for x in (self.mean[i], self.variance[i], self.minimum[i],
                                                                                              self.maximum[i], self.ci_lower[i], self.ci_upper[i])]))
        return "\n".join(rows) + "\n"


def summarize_node_ages(ages: np.ndarray, burn_in: float = 0.25, names: Optional[Sequence[str]] = None) -> AgeSummary:
    """ages: [n_samples, n_nodes] of one chain (or of pooled chains, samples along axis 0).  The first
    round(n_samples * burn_in) samples are dropped (`scripts/analyze` passes 0.25 after its own thinning)."""
    a = np.asarray(ages, dtype=np.float64)
    if a.ndim != 2:
        raise ValueError("summarize_node_ages: expected [n_samples, n_nodes]")
    l0 = a.shape[0]
    a = a[int(round(l0 * burn_in)):]
    l = a.shape[0]
    if l == 0:
        raise ValueError("summarize_node_ages: no samples left after burn-in")
    mean = a.mean(axis=0)
    var = a.var(axis=0)                                   # statistics' meanVariance: maximum-likelihood estimate
    srt = np.sort(a, axis=0)
    i_ci = int(math.floor(l * 0.025))
    n_ci = int(math.floor(l * 0.95))
    if n_ci < 1:
        raise ValueError("summarize_node_ages: too few samples for the 95 % interval")
    n = a.shape[1]
    return AgeSummary(np.arange(n), list(names) if names is not None else [""] * n, mean, var, srt[0], srt[-1], srt[i_ci], srt[i_ci + n_ci - 1])
