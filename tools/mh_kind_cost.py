#!/usr/bin/env python3
"""Per-proposal-kind cost of one lock step of the chain kernel (diagnostic): runs a schedule made of a single kind."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import mcmc_date_amd as M
    from test_gpu_mh import setup

    name = sys.argv[1] if len(sys.argv) > 1 else "12-leaves-variable-rate"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    fx = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
    topo, ps, smp, _ = setup(fx, B=B, seed=1)
    smp.run(30)
    smp.autotune()
    kinds = sorted({p.kind for p in ps})
    for k in kinds:
        rows = np.array([i for i, p in enumerate(ps) if p.kind == k], np.int32)
        sched = np.resize(rows, (1, 4000))
        smp.run_schedule(sched[:, :400])
        t0 = time.perf_counter()
        smp.run_schedule(sched)
        dt = time.perf_counter() - t0
        print(f"kind {k:2d}  {1e6 * dt / sched.size:7.2f} us per lock step  ({len(rows)} rows)", flush=True)


if __name__ == "__main__":
    main()
