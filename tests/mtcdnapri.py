"""The reference's own 7-taxon mtCDNApri analysis (bench/comparison_with_mcmctree/README.md:615-632) set up from the committed
golden inputs -- shared by the CPU tests of the twin (tests/test_reference_samples.py) and the GPU tests of the device sampler
(tests/test_gpu_mh.py).  Nothing here touches /root/reference: the three input files are inside
tests/golden/mtCDNApri_prior_samples.json.

ROOT_UPPER_OF_THE_SAMPLES: the committed calibration tree bounds the root by 'U(100,2.5e-2)', the committed samples were drawn
with a soft upper bound of 30.0 (tail mass 0.025) -- they stop at 31.5 in all six prior-only runs, at 30.9 in the posterior
runs, and the edge fit of test_reference_samples.py gives 30.0 +- 0.1 with the tail width of calibrateSoftF
(lib/Mcmc/Tree/Prior/Node/Calibration.hs:369-391).  `analysis(root_upper=...)` replaces the bound."""
import dataclasses
import json
import os
import tempfile

import numpy as np

import mcmc_date_amd as M
from mcmc_date_amd.prepare import prepare

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT_UPPER_OF_THE_SAMPLES = 30.0


def golden(which):
    return json.load(open(os.path.join(GOLDEN, f"mtCDNApri_{which}_samples.json")))


@dataclasses.dataclass
class Analysis:
    prep: object
    topo: object
    cal: list
    ht: float
    table: list
    mu: np.ndarray
    sigma_inv: np.ndarray          # dense (the twin's operand; NoData: 1e-12 I, flat over the whole support)
    logdet: float


def analysis(likelihood_spec="NoLikelihood", root_upper=None, exact_jacobians=False) -> Analysis:
    """`./run -s -f analysis.conf -c ul {n|s} p` + the run-time set-up of app/Main.hs:370-457: prepare, calibrations from the
    MCMCtree-style tree, ht = getMeanRootHeight, the proposal cycle with calibrations available."""
    fx = golden("prior")
    with tempfile.TemporaryDirectory() as d:
        paths = {}
        for k in ("rooted_tree", "calibration_tree", "tree_list"):
            paths[k] = os.path.join(d, k)
            open(paths[k], "w").write(fx["inputs"][k])
        prep = prepare(paths["tree_list"], paths["rooted_tree"], likelihood_spec)
        topo = prep.topology
        cal = M.load_calibrations_from_tree(topo, paths["calibration_tree"])
    if root_upper is not None:
        assert [c.node for c in cal].count(0) == 1
        cal = [dataclasses.replace(c, upper=float(root_upper)) if c.node == 0 else c for c in cal]
    ht = M.get_mean_root_height(cal)
    ps, missing = M.proposals(topo, [], calibrations_available=True, exact_jacobians=exact_jacobians)
    assert missing == []
    n = topo.n_nodes - 2
    if likelihood_spec == "NoLikelihood":
        mu, P, logdet = np.full(n, 0.5), np.eye(n) * 1e-12, 0.0          # NoData: likelihood 1 (app/Probability.hs:281)
    else:
        mu, logdet = np.asarray(prep.mu, float), float(prep.lhd.logdet_sigma)
        P = np.zeros((n, n))
        for (i, j), v in prep.lhd.sigma_inv_assoc:
            P[i, j] = v
    return Analysis(prep, topo, cal, ht, ps, mu, P, logdet)


def twin_chains(an: Analysis, B, seed, start_height=1.0):
    """oracle.MhChains for the analysis, every chain at initWith (time height 1.0 as the reference starts)."""
    import oracle as O

    spec = O.PriorSpec(an.topo.parent, an.ht, "UncorrelatedLogNormal",
                       [(c.node, c.lower, c.lower_p, c.upper, c.upper_p) for c in an.cal], [], [])
    model = O.MhModel(an.topo.parent, an.mu, an.sigma_inv, an.logdet, spec, M.table_arrays(an.table))
    x0 = M.init_with(an.topo, an.prep.mean_lengths)
    x0.time_height = start_height
    s0 = M.StateBatch.from_states([x0] * B)
    return O.MhChains(model, s0.time_birth_rate, s0.time_death_rate, s0.time_height, s0.heights, s0.rate_mean, s0.rate_variance,
                      s0.rates, seed=seed)


def monitored_ages(chains, table, seed, iterations=M.sampler.ITERATIONS, period=2, burn_in_fraction=0.25):
    """What the reference's `prior_samples_run*.tsv` / `post_samples_run*.tsv` hold: the absolute node ages monitored every 2
    iterations over burn-in (with auto tuning, app/Definitions.hs:420-424) AND the 8000 iterations that follow, the first 25 % of
    the monitored lines dropped by the summary script (scripts/analyze:38: `trees-monitor-summary-ultrametric FILE 0.25`; 6466 lines -> the 4850 rows of a run).  `chains`: the
    CPU twin (oracle.MhChains) or anything with run(schedule) / autotune() / tH / H.  Returns [rows, chains, n_nodes]."""
    rng = np.random.default_rng([int(seed), 0xA6E5])
    out = [chains.tH[:, None] * chains.H]                            # the monitor's line of iteration 0
    for p in list(M.sampler.BURN_IN_FAST) + list(M.sampler.BURN_IN_SLOW):
        done = 0
        while done < p:
            k = min(period, p - done)
            chains.run(M.cycle_schedule(table, k, rng))
            done += k
            out.append(chains.tH[:, None] * chains.H)
        chains.autotune()
    for _ in range(iterations // period):
        chains.run(M.cycle_schedule(table, period, rng))
        out.append(chains.tH[:, None] * chains.H)
    a = np.array(out)
    return a[int(round(a.shape[0] * burn_in_fraction)):]
