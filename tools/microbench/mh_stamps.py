"""Phase stamps of the Metropolis-Hastings chain kernel (cycles per step: loop head, propose, prior, likelihood, accept).
Build the stamped library first:  make -C mcmc-date_amd/csrc stamp_mh   (writes tools/microbench/libmhstamp.so),
then on the GPU box:  python tools/microbench/mh_stamps.py [chains]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MCD_LIB_PATH"] = os.path.join(ROOT, "tools", "microbench", "libmhstamp.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mcmc_date_amd as M
from test_gpu_mh import setup
fx = dict(np.load(os.path.join(ROOT, "tests", "golden", "12-leaves-variable-rate.npz")))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
topo, ps, smp, _ = setup(fx, B=B, seed=1)
smp.run(30); smp.autotune()
sched = M.cycle_schedule(ps, 10, np.random.default_rng(0))
ta, tk = smp.run_schedule(sched, trace=True)
cyc = ta[:5].mean(axis=1) / sched.size
print("cycles per step: head %.0f propose %.0f prior %.0f likelihood %.0f accept %.0f  total %.0f" % (*cyc, cyc.sum()))
