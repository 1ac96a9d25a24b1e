"""Where does a long run of the small-tree kernel stop?  The posterior test's device side (burn_in + run) with a line per launch.
On the GPU box:  timeout -k 5 150 python tools/debug/chain_spec_debug.py [thread]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mcmc_date_amd as M
import test_gpu_mh as T
g = np.load(os.path.join(ROOT, "tests", "golden", "12-leaves-variable-rate.npz"), allow_pickle=True)
fx = {k: g[k] for k in g.files}
topo, ps, smp, _ = T.setup(fx, B=64, seed=1001)
_, _, _, twin = T.setup(fx, B=32, seed=2002)
t0 = time.time()
orig = smp.run_schedule
n = [0]
def traced(sched, **kw):
    n[0] += 1
    print("launch", n[0], "steps", np.asarray(sched).size, end=" ... ", flush=True)
    r = orig(sched, **kw)
    smp.posterior()
    print("done", round(time.time() - t0, 2), flush=True)
    return r
smp.run_schedule = traced
if len(sys.argv) > 1:
    def cpu_side():
        rng = np.random.default_rng(9)
        for period in M.sampler.BURN_IN_FAST[:6]:
            twin.run(M.cycle_schedule(ps, period, rng)); twin.autotune()
    th = threading.Thread(target=cpu_side); th.start()
smp.burn_in()
print("burn-in done", flush=True)
smp.run(512, accumulate=True)
print("run done", flush=True)
