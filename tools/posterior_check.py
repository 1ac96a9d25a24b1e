#!/usr/bin/env python3
"""Posterior-level check on any golden dataset: device sampler (64 chains) against the CPU twin (32 chains, different
seed), reference burn-in and N iterations; prints the largest relative difference of the inner nodes' age means.
Usage: python tools/posterior_check.py <dataset> [iterations=4000]"""
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import mcmc_date_amd as M
    from test_gpu_mh import setup

    name = sys.argv[1]
    n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    fx = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
    topo, ps, smp, _ = setup(fx, B=64, seed=1001)
    _, _, _, twin = setup(fx, B=32, seed=2002)

    def cpu_side():
        rng = np.random.default_rng(9)
        for period in M.sampler.BURN_IN_FAST + M.sampler.BURN_IN_SLOW:
            twin.run(M.cycle_schedule(ps, period, rng))
            twin.autotune()
        twin.run(M.cycle_schedule(ps, n_iter, rng), accumulate=True)

    th = threading.Thread(target=cpu_side)
    th.start()
    smp.burn_in()
    smp.run(n_iter, accumulate=True)
    th.join()
    mean_gpu = smp.node_age_summary()[0]
    mean_cpu = twin.age_sum.mean(axis=0) / twin.n_samples
    inner = ~topo.leaves
    rel = np.abs(mean_gpu[inner] - mean_cpu[inner]) / mean_cpu[inner]
    print(json.dumps({"dataset": name, "n_nodes": topo.n_nodes, "proposals": len(ps), "steps_per_iteration": int(sum(p.weight for p in ps)),
                      "iterations": n_iter, "max_rel_diff_node_age_means": float(rel.max()), "median_rel_diff": float(np.median(rel))}))


if __name__ == "__main__":
    main()
