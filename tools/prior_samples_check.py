#!/usr/bin/env python3
"""Device sampler WITHOUT data on the reference's own mtCDNApri prior-only analysis, against the summary statistics of the six
prior-only chains the reference ships (tests/golden/mtCDNApri_prior_samples.json; generator make_prior_sample_summary.py).
Prints one JSON line per Jacobian variant.  Usage: python tools/prior_samples_check.py [chains=256] [period=10]"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(exact_jacobians, B, period, seed, fx, lift=1, start_height=None, tune=True):
    import mcmc_date_amd as M
    from mcmc_date_amd import monitor as MO
    from mcmc_date_amd.prepare import prepare

    with tempfile.TemporaryDirectory() as d:
        paths = {}
        for k in ("rooted_tree", "calibration_tree", "tree_list"):
            paths[k] = os.path.join(d, k)
            open(paths[k], "w").write(fx["inputs"][k])
        prep = prepare(paths["tree_list"], paths["rooted_tree"], "NoLikelihood")      # `run ... n p`
        topo = prep.topology
        cal = M.load_calibrations_from_tree(topo, paths["calibration_tree"])
    ht = M.get_mean_root_height(cal)
    n = topo.n_nodes - 2
    # NoData: likelihood 1 (app/Probability.hs:281); here a precision matrix of 1e-12 I, flat over the whole support
    lik = M.MvnLikelihood(M.Full(np.full(n, 0.5), np.eye(n) * 1e-12, 0.0)).bind_tree(topo)
    pf = M.PriorFunction(ht, "UncorrelatedLogNormal", cal, [], [], topo)
    ps, missing = M.proposals(topo, [], calibrations_available=True, exact_jacobians=exact_jacobians)
    assert missing == []
    import dataclasses
    if lift == "all":                                    # every proposal lifted: the chain targets prior x jacobianRootBranch exactly
        ps = [dataclasses.replace(p, jac_root=1) for p in ps]
    else:
        ps = [dataclasses.replace(p, jac_root=int(p.jac_root) * lift) for p in ps]   # lift: 1 as restated, 0 no root-branch lift, -1 reciprocal
    smp = M.Sampler(lik, pf, ps, B, seed=seed)
    x0 = M.init_with(topo, prep.mean_lengths)
    x0.time_height = ht if start_height is None else start_height
    smp.set_initial_state(x0)
    if tune:
        smp.burn_in()                               # burnIn, app/Definitions.hs:420-424
    else:
        smp.run(4930)                               # the same number of iterations, the proposals left at their initial sizes
    tr = MO.collect(smp, 8000, period=period)       # iterations, :440-441
    ages = tr.ages()[:, :, fx["nodes"]].reshape(-1, len(fx["nodes"]))
    q = np.quantile(ages, [0.025, 0.5, 0.975], axis=0)
    chain_means = tr.ages()[:, :, fx["nodes"]].mean(axis=0)
    return {"exact_jacobians": exact_jacobians, "lift": lift, "chains": B, "samples": int(ages.shape[0]), "ht": ht,
            "calibrations": [(c.name, c.node, c.lower, c.lower_p, c.upper, c.upper_p) for c in cal],
            "mean": ages.mean(axis=0).tolist(), "sd": ages.std(axis=0, ddof=1).tolist(), "q025": q[0].tolist(), "q50": q[1].tolist(),
            "q975": q[2].tolist(), "se_mean": (chain_means.std(axis=0, ddof=1) / np.sqrt(B)).tolist()}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    period = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "mtCDNApri_prior_samples.json")))
    ref = fx["pooled"]
    print(json.dumps({"reference": {k: np.round(ref[k], 4).tolist() for k in ("mean", "sd", "q025", "q50", "q975")},
                      "between_run_sd_of_mean": fx["between_run_sd_of_mean"]}))
    # (exact_jacobians, lift, starting time height, auto tuning); `all`: the experiments recorded in profiles/r02_prior_samples_variants.jsonl
    variants = ([(False, 1, None, True), (True, 1, None, True), (False, 0, None, True), (False, -1, None, True), (False, "all", None, True),
                 (False, 2, None, True), (False, 1, 1.0, True), (False, 1, 20.0, False)] if len(sys.argv) > 3
                else [(False, 1, None, True), (True, 1, None, True)])
    for ej, lift, sh, tune in variants:
        r = run(ej, B, period, 11 + int(ej), fx, lift, sh, tune)
        r["start_height"] = sh
        r["tuned"] = tune
        r["mean_minus_ref_in_ref_se"] = ((np.array(r["mean"]) - np.array(ref["mean"])) / (np.array(fx["between_run_sd_of_mean"]) / np.sqrt(6))).round(2).tolist()
        r["rel_dev_mean"] = ((np.array(r["mean"]) - np.array(ref["mean"])) / np.array(ref["mean"])).round(4).tolist()
        print(json.dumps(r))


if __name__ == "__main__":
    main()
