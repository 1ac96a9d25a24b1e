#!/usr/bin/env python3
"""MH proposal-steps per second of the lock-step driver (mcd_mh_*) on a golden dataset, next to the CPU twin
(oracle/mh_oracle.c, OpenMP over chains).  BASELINE.json configs[1]: tests/12-leaves-variable-rate, 64 chains.
One JSON line per chain count.  Usage: python tools/bench_mh.py [--name 12-leaves-variable-rate] [--chains 64,4096]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--name", default="12-leaves-variable-rate")
    ap.add_argument("--chains", default="64,4096")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--cpu-iters", type=int, default=20)
    args = ap.parse_args()
    import mcmc_date_amd as M
    import oracle as O
    from test_gpu_mh import setup

    fx = dict(np.load(os.path.join(ROOT, "tests", "golden", args.name + ".npz")))
    for B in [int(x) for x in args.chains.split(",")]:
        topo, ps, smp, twin = setup(fx, B=B, seed=1)
        S = sum(p.weight for p in ps)
        smp.run(20)
        smp.autotune()
        t0 = time.perf_counter()
        smp.run(args.iters)
        dt = time.perf_counter() - t0
        rng = np.random.default_rng(0)
        twin.run(M.cycle_schedule(ps, 2, rng))
        t1 = time.perf_counter()
        twin.run(M.cycle_schedule(ps, args.cpu_iters, rng))
        dc = time.perf_counter() - t1
        print(json.dumps({"metric": "MH proposal steps/sec (chains x steps)", "dataset": args.name, "chains": B, "n_nodes": topo.n_nodes,
                          "steps_per_iteration": S, "gpu_steps_per_s": B * S * args.iters / dt, "gpu_us_per_lockstep": 1e6 * dt / (S * args.iters),
                          "cpu_twin_steps_per_s": B * S * args.cpu_iters / dc, "cpu_threads": os.cpu_count()}), flush=True)


if __name__ == "__main__":
    main()
