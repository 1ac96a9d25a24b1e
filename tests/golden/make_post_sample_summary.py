#!/usr/bin/env python3
"""Build tests/golden/mtCDNApri_post_samples.json from the reference's own POSTERIOR runs (build container only).

bench/comparison_with_mcmctree/03_compare_estimates/post_samples_run{1..6}.tsv: node ages sampled by six independent
`mcmc-date-run run` chains WITH data on the 7-taxon mtCDNApri analysis (`./run -s -f analysis.conf -c ul s p` then `... s r`,
README.md:615-632): calibrations from the MCMCtree-style tree, uncorrelated log-normal clock, likelihood
`SparseMultivariateNormal 0.1` (scripts/run:137: the precision matrix is the graphical-lasso estimate with penalty 0.1 on the
correlation matrix of the ten PhyloBayes trees, app/Main.hs:257-276).  Same layout as the prior-only samples (see
make_prior_sample_summary.py, whose JSON also holds the three input files).  These are outputs of the reference itself that
depend on the whole path: prepare, likelihood, prior, proposal cycle, Jacobians.

Committed: summary statistics only (data, not source)."""
import json
import os

import numpy as np

REF = "/root/reference/bench/comparison_with_mcmctree"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    runs = []
    for i in range(1, 7):
        a = np.loadtxt(os.path.join(REF, "03_compare_estimates", f"post_samples_run{i}.tsv"), skiprows=1)
        runs.append(a[:, 1:])
    header = open(os.path.join(REF, "03_compare_estimates", "post_samples_run1.tsv")).readline().split()[1:]
    nodes = [int(x) for x in header]
    allr = np.concatenate(runs)

    def stats(a):
        q = np.quantile(a, [0.025, 0.5, 0.975], axis=0)
        return {"mean": a.mean(axis=0).tolist(), "sd": a.std(axis=0, ddof=1).tolist(), "q025": q[0].tolist(), "q50": q[1].tolist(),
                "q975": q[2].tolist()}

    run_means = np.array([r.mean(axis=0) for r in runs])
    out = {
        "source": "dschrempf/mcmc-date bench/comparison_with_mcmctree/03_compare_estimates/post_samples_run{1..6}.tsv (README.md:615-632)",
        "command": "./run -s -f analysis.conf -c ul s r   (calibrations, UncorrelatedLogNormal, SparseMultivariateNormal 0.1)",
        "nodes": nodes, "rows_per_run": [int(r.shape[0]) for r in runs], "pooled": stats(allr), "runs": [stats(r) for r in runs],
        "between_run_sd_of_mean": run_means.std(axis=0, ddof=1).tolist(),
        "correlation": np.corrcoef(allr.T).tolist(),
        "root_age_max_per_run": [float(r[:, 0].max()) for r in runs],
    }
    with open(os.path.join(HERE, "mtCDNApri_post_samples.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("nodes", nodes, "rows", out["rows_per_run"])
    print("pooled mean", np.round(out["pooled"]["mean"], 3))
    print("between-run sd of the mean", np.round(out["between_run_sd_of_mean"], 3))
    print("pooled q025", np.round(out["pooled"]["q025"], 3), "q975", np.round(out["pooled"]["q975"], 3))


if __name__ == "__main__":
    main()
