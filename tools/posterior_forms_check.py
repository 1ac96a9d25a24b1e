#!/usr/bin/env python3
"""End-to-end check of the two log-density forms under the Metropolis-Hastings driver: the same chains (same seeds, same
schedule) on a synthetic large tree, once with the column sweeps only and once with the automatic choice (multiply form
for this batch).  The random streams are counter based, so the two runs propose the same moves; they can only part ways
where the last bits of ln alpha decide an accept.  Prints how many chains end in a different state and how far the
pooled node-age means are apart.
Usage: python tools/posterior_forms_check.py [n_leaves=128] [chains=4096] [iterations=20]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(form, n_leaves, B, n_iter):
    import mcmc_date_amd as M
    from mcmc_date_amd import synthetic as S

    M.set_logpdf_form(form)
    topo = S.random_topology(n_leaves, seed=3)
    n = topo.n_nodes - 2
    mu, sigma = S.random_spd_problem(n, seed=3)
    lik = M.MvnLikelihood.from_covariance(mu, sigma).bind_tree(topo)
    pf = M.PriorFunction(1.0, "UncorrelatedGamma", [], [], [], topo)
    ps, _ = M.proposals(topo, [], calibrations_available=True)
    s0 = S.random_states(topo, B, seed=4)
    s0.time_birth_rate = np.full(B, 1.0); s0.time_death_rate = np.full(B, 0.8); s0.rate_variance = np.full(B, 0.3)
    smp = M.Sampler(lik, pf, ps, B, seed=13)
    smp.set_state(s0)
    t0 = time.perf_counter()
    smp.run(n_iter // 2)
    smp.autotune()
    smp.run(n_iter - n_iter // 2, accumulate=True)
    dt = time.perf_counter() - t0
    st = smp.state()
    mean = smp.node_age_summary()[0]
    M.set_logpdf_form("auto")
    return st, mean, dt, topo, int(sum(p.weight for p in ps))


def main():
    n_leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    n_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    st_s, mean_s, dt_s, topo, w = run("sweep", n_leaves, B, n_iter)
    st_a, mean_a, dt_a, _, _ = run("auto", n_leaves, B, n_iter)
    same = np.all(st_s.heights == st_a.heights, axis=1) & np.all(st_s.rates == st_a.rates, axis=1)
    close = np.all(np.abs(st_s.heights - st_a.heights) <= 1e-9, axis=1) & np.all(np.abs(st_s.rates - st_a.rates) <= 1e-9 * np.abs(st_s.rates), axis=1)
    inner = ~topo.leaves
    rel = np.abs(mean_s[inner] - mean_a[inner]) / mean_s[inner]
    print(json.dumps({"n_nodes": topo.n_nodes, "chains": B, "iterations": n_iter, "lock_steps": n_iter * w,
                      "chains_bit_identical_at_the_end": int(same.sum()), "chains_within_1e-9": int(close.sum()),
                      "max_rel_diff_pooled_node_age_means": float(rel.max()), "seconds_sweep": round(dt_s, 2), "seconds_auto": round(dt_a, 2)}))


if __name__ == "__main__":
    main()
