#!/usr/bin/env python3
"""Device leapfrog (prior + likelihood gradient of the Hamiltonian target) on a synthetic tree: microseconds per leapfrog
step for all chains of the batch.  Usage: python tools/bench_leapfrog.py [n_leaves=128] [chains=4096] [steps=200]
(MCD_WIDE=0 in the environment pins the column sweeps for comparison.)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import mcmc_date_amd as M
    from mcmc_date_amd import synthetic as S

    n_leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    topo = S.random_topology(n_leaves, seed=3)
    n = topo.n_nodes - 2
    mu, sigma = S.random_spd_problem(n, seed=3)
    lik = M.MvnLikelihood.from_covariance(mu, sigma).bind_tree(topo)
    pf = M.PriorFunction(1.0, "UncorrelatedGamma", [], [], [], topo)
    s0 = S.random_states(topo, B, seed=4)
    s0.time_birth_rate = np.full(B, 1.0); s0.time_death_rate = np.full(B, 0.8); s0.rate_variance = np.full(B, 0.3)
    lf = M.Leapfrog(lik, pf, True, B)
    lf.set_state(s0)
    rng = np.random.default_rng(1)
    p0 = rng.standard_normal((B, lf.dim))
    inv_mass = np.ones(lf.dim)
    lf.leapfrog(p0, 1e-4, inv_mass, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lf.leapfrog(p0, 1e-4, inv_mass, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "leapfrog step, all chains", "n_nodes": topo.n_nodes, "chains": B, "steps": steps, "us_per_step": 1e6 * dt / steps,
                      "chain_steps_per_s": B * steps / dt, "form": os.environ.get("MCD_WIDE", "auto")}))


if __name__ == "__main__":
    main()
