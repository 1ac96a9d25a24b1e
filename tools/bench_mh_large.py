#!/usr/bin/env python3
"""Lock-step Metropolis-Hastings on a synthetic large tree (two launches per step): microseconds per lock step.
Usage: python tools/bench_mh_large.py [n_leaves=128] [chains=512] [steps=2000] [dense|sparse]
`sparse`: the precision matrix kept sparse on the device (mcd_mh_create_sparse; trees up to 2048 nodes)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import mcmc_date_amd as M
    from mcmc_date_amd import synthetic as S

    n_leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    topo = S.random_topology(n_leaves, seed=3)
    n = topo.n_nodes - 2
    form = sys.argv[4] if len(sys.argv) > 4 else "dense"
    if form == "sparse":
        _, assoc = S.banded_precision(n, seed=3)
        mu = np.random.default_rng(3).uniform(0.01, 0.2, n)
        lik = M.SparseLikelihood(M.Sparse(mu, assoc, 0.0)).bind_tree(topo)
    else:
        mu, sigma = S.random_spd_problem(n, seed=3)
        lik = M.MvnLikelihood.from_covariance(mu, sigma).bind_tree(topo)
    cal, con = [], []
    if os.environ.get("MCD_BENCH_CAL"):                      # some node priors: calibrations on internal nodes, constraints between a node and an ancestor
        k = int(os.environ["MCD_BENCH_CAL"])
        inner = [v for v in range(1, topo.n_nodes) if (np.asarray(topo.parent) == v).any()]
        rng = np.random.default_rng(5)
        for i, v in enumerate(rng.choice(inner, size=min(k, len(inner)), replace=False)):
            cal.append(M.Calibration(f"c{i}", int(v), 1e-3, 0.025, 10.0, 0.025))
        for i, v in enumerate(rng.choice(inner, size=min(k // 2, len(inner)), replace=False)):
            a = int(topo.parent[int(v)])
            if a > 0:
                con.append(M.Constraint(f"k{i}", int(v), a, 0.025))
    pf = M.PriorFunction(1.0, "UncorrelatedGamma", cal, con, [], topo)
    ps, _ = M.proposals(topo, [], calibrations_available=True)
    s0 = S.random_states(topo, B, seed=4)
    s0.time_birth_rate = np.full(B, 1.0); s0.time_death_rate = np.full(B, 0.8); s0.rate_variance = np.full(B, 0.3)
    smp = M.Sampler(lik, pf, ps, B, seed=13)
    smp.set_state(s0)
    sched = M.cycle_schedule(ps, 1, np.random.default_rng(0))
    reps = max(1, steps // sched.shape[1] + 1)
    sched = np.tile(sched, (1, reps))[:, :steps]
    smp.run_schedule(sched[:, :200])
    t0 = time.perf_counter()
    smp.run_schedule(sched)
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "MH lock step, large tree", "n_nodes": topo.n_nodes, "chains": B, "steps": steps,
                      "us_per_lockstep": 1e6 * dt / steps, "steps_per_s": B * steps / dt, "proposals_per_iteration": int(sum(p.weight for p in ps)), "form": form, "path": smp.last_path()}))


if __name__ == "__main__":
    main()
