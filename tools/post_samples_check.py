#!/usr/bin/env python3
"""Device sampler WITH data on the reference's own mtCDNApri posterior analysis (likelihood `SparseMultivariateNormal 0.1`:
graphical lasso in `prepare`), against the summary statistics of the six posterior chains the reference ships
(tests/golden/mtCDNApri_post_samples.json; inputs in mtCDNApri_prior_samples.json).  One JSON line per variant of the
graphical lasso's convention (diagonal penalised or not) and of the Jacobians.
Usage: python tools/post_samples_check.py [chains=256] [period=10]"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(pen_diag, exact_jacobians, B, period, seed, fx, nodes, spec="SparseMultivariateNormal 0.1"):
    import mcmc_date_amd as M
    from mcmc_date_amd import monitor as MO
    from mcmc_date_amd import prepare as P

    P.GLASSO_PENALIZE_DIAGONAL = pen_diag
    with tempfile.TemporaryDirectory() as d:
        paths = {}
        for k in ("rooted_tree", "calibration_tree", "tree_list"):
            paths[k] = os.path.join(d, k)
            open(paths[k], "w").write(fx["inputs"][k])
        prep = P.prepare(paths["tree_list"], paths["rooted_tree"], spec)                # `run ... s p`
        topo = prep.topology
        cal = M.load_calibrations_from_tree(topo, paths["calibration_tree"])
    ht = M.get_mean_root_height(cal)
    lik = M.MvnLikelihood(prep.lhd).bind_tree(topo)
    pf = M.PriorFunction(ht, "UncorrelatedLogNormal", cal, [], [], topo)
    ps, missing = M.proposals(topo, [], calibrations_available=True, exact_jacobians=exact_jacobians)
    assert missing == []
    smp = M.Sampler(lik, pf, ps, B, seed=seed)
    x0 = M.init_with(topo, prep.mean_lengths)
    x0.time_height = ht
    smp.set_initial_state(x0)
    smp.burn_in()
    tr = MO.collect(smp, 8000, period=period)
    ages = tr.ages()[:, :, nodes].reshape(-1, len(nodes))
    q = np.quantile(ages, [0.025, 0.5, 0.975], axis=0)
    nz = len(prep.lhd.sigma_inv_assoc) if hasattr(prep.lhd, "sigma_inv_assoc") else -1
    return {"likelihood": spec, "penalize_diagonal": pen_diag, "exact_jacobians": exact_jacobians, "chains": B, "samples": int(ages.shape[0]),
            "precision_entries": nz, "mean": ages.mean(axis=0).tolist(), "sd": ages.std(axis=0, ddof=1).tolist(), "q025": q[0].tolist(),
            "q50": q[1].tolist(), "q975": q[2].tolist()}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    period = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "mtCDNApri_prior_samples.json")))
    post = json.load(open(os.path.join(ROOT, "tests", "golden", "mtCDNApri_post_samples.json")))
    ref = post["pooled"]
    print(json.dumps({"reference": {k: np.round(ref[k], 4).tolist() for k in ("mean", "sd", "q025", "q50", "q975")},
                      "between_run_sd_of_mean": post["between_run_sd_of_mean"]}))
    for pen, ej in ((True, False), (False, False), (True, True), (False, True)):
        r = run(pen, ej, B, period, 21 + int(ej), fx, post["nodes"])
        r["rel_dev_mean"] = ((np.array(r["mean"]) - np.array(ref["mean"])) / np.array(ref["mean"])).round(4).tolist()
        r["rel_dev_q025"] = ((np.array(r["q025"]) - np.array(ref["q025"])) / np.array(ref["q025"])).round(4).tolist()
        r["rel_dev_q975"] = ((np.array(r["q975"]) - np.array(ref["q975"])) / np.array(ref["q975"])).round(4).tolist()
        print(json.dumps(r))


if __name__ == "__main__":
    main()
