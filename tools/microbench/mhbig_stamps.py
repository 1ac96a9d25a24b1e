"""Phase ticks of the streaming chain kernel (csrc/k_mh_chain_big.hip), per chain wave, summed over a run: share of
loop head | propose | prior | distances | first ring barrier + sweep | accept.
Build first:  make -C mcmc-date_amd/csrc stamp_mhbig ;  on the GPU box:  python tools/microbench/mhbig_stamps.py [n] [chains] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MCD_LIB_PATH"] = os.path.join(ROOT, "tools", "microbench", os.environ.get("MHBSTAMPLIB", "libmhbigstamp.so"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mcmc_date_amd as M
from mcmc_date_amd import synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
topo = S.random_topology((n + 3) // 2, seed=3)
nd = topo.n_nodes - 2
mu, sigma = S.random_spd_problem(nd, seed=3)
tl = M.MvnLikelihood.from_covariance(mu, sigma).bind_tree(topo)
pf = M.PriorFunction(1.0, "UncorrelatedGamma", [], [], [], topo)
ps, _ = M.proposals(topo, [], calibrations_available=True)
s0 = S.random_states(topo, B, seed=4)
s0.time_birth_rate = np.full(B, 1.0); s0.time_death_rate = np.full(B, 0.8); s0.rate_variance = np.full(B, 0.3)
smp = M.Sampler(tl, pf, ps, B, seed=13)
smp.set_state(s0)
cyc = M.cycle_schedule(ps, 1, np.random.default_rng(0))
sched = np.tile(cyc, (1, steps // cyc.shape[1] + 1))[:, :steps]
if len(sys.argv) > 4:
    # one SEGMENT of a large tree's schedule (k_mh_chain_big<R, true>): only proposals of the given kinds, at most 250 steps (one launch)
    kinds = [int(k) for k in sys.argv[4].split(",")]
    tab = M.table_arrays(ps)
    keep = np.isin(tab["kind"][cyc[0]], kinds)
    steps = min(steps, 250)
    sched = cyc[:, keep][:, :steps]
    steps = sched.shape[1]
smp.run_schedule(sched[:, :200])
t0 = time.perf_counter()
ta, _ = smp.run_schedule(sched, trace=True)
dt = time.perf_counter() - t0
tk = ta[:10].mean(axis=1)
n_sparse = tk[7]
tk = np.concatenate([tk[:7], tk[8:10]])
print("us per lock step %.2f (n_nodes %d, chains %d; with tracing)" % (1e6 * dt / steps, topo.n_nodes, B))
names = ["loop head", "propose", "prior", "distances + column requests (R <= 4: the request to the likelihood wave)", "barrier + sweep (dense steps)", "accept", "column update (sparse steps; R <= 4: waiting for |z'|^2)", "draws of 64 steps (per step)", "distances + ln Jacobian (R <= 4: dense steps only)"]
print("  sparse steps: %.1f %% of %d" % (100 * n_sparse / steps, steps))
for nm, v in zip(names, tk):
    print("  %-60s %5.1f %%   (%.0f ticks per step)" % (nm, 100 * v / tk.sum(), v / steps))
