"""Phase ticks of the segment kernel's chain wave (csrc/k_mh_segment.hip), summed over ONE segment of at most 250 steps made of the
given proposal kinds (MCD_PROP_* numbers, comma separated; default: every kind a segment may hold).
Build first:  make -C mcmc-date_amd/csrc stamp_seg ;  on the GPU box:  python tools/microbench/seg_stamps.py [n] [chains] [kinds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MCD_LIB_PATH"] = os.path.join(ROOT, "tools", "microbench", os.environ.get("SEGSTAMPLIB", "libsegstamp.so"))
sys.path.insert(0, ROOT)
import numpy as np
import mcmc_date_amd as M
from mcmc_date_amd import synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
kinds = [int(k) for k in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,5,10,11").split(",")]
sparse = len(sys.argv) > 4 and sys.argv[4] == "sparse"      # the sparse driver's segment kernel (k_mh_segment_sparse.hip: the same chain wave)
topo = S.random_topology((n + 3) // 2, seed=3)
nd = topo.n_nodes - 2
if sparse:
    _, assoc = S.banded_precision(nd, seed=3)
    tl = M.SparseLikelihood(M.Sparse(np.random.default_rng(3).uniform(0.01, 0.2, nd), assoc, 0.0)).bind_tree(topo)
else:
    mu, sigma = S.random_spd_problem(nd, seed=3)
    tl = M.MvnLikelihood.from_covariance(mu, sigma).bind_tree(topo)
pf = M.PriorFunction(1.0, "UncorrelatedGamma", [], [], [], topo)
ps, _ = M.proposals(topo, [], calibrations_available=True)
s0 = S.random_states(topo, B, seed=4)
s0.time_birth_rate = np.full(B, 1.0); s0.time_death_rate = np.full(B, 0.8); s0.rate_variance = np.full(B, 0.3)
smp = M.Sampler(tl, pf, ps, B, seed=13)
smp.set_state(s0)
cyc = M.cycle_schedule(ps, 1, np.random.default_rng(0))
tab = M.table_arrays(ps)
size = np.ones(topo.n_nodes, int)
for v in range(topo.n_nodes - 1, 0, -1):
    size[topo.parent[v]] += size[v]
k_of = tab["kind"][cyc[0]]
keep = np.isin(k_of, kinds) & ~(np.isin(k_of, [2, 5, 11]) & (size[tab["node"][cyc[0]]] > 60))      # (sub trees a segment surely holds)
sched = cyc[:, keep][:, :250]
steps = sched.shape[1]
smp.run_schedule(sched)
t0 = time.perf_counter()
ta, _ = smp.run_schedule(sched, trace=True)
dt = time.perf_counter() - t0
assert "segments" in smp.last_path(), smp.last_path()
tk = ta[:8].mean(axis=1)
print("us per lock step %.2f (n_nodes %d, chains %d, %d steps of kinds %s in one launch; with tracing)" % (1e6 * dt / steps, topo.n_nodes, B, steps, kinds))
for nm, v in zip(["loop head + draws", "propose", "posting the transform", "ln prior", "waiting for |z'|^2 / q'", "decision + commit", "(start) waiting for the prior waves' done", "(start) applying the transform"], tk):
    print("  %-26s %5.1f %%   (%.0f ticks per step)" % (nm, 100 * v / tk.sum(), v / steps))
if sparse:
    lk = ta[8:13].mean(axis=1)
    print("  likelihood wave (sparse):")
    for nm, v in zip(["waiting for the request", "ahead of it: list from the row, fetches", "request -> deltas, look-ups", "sum + answer posted", "waiting for the decision + commit"], lk):
        print("    %-36s %5.1f %%   (%.0f ticks per step)" % (nm, 100 * v / lk.sum(), v / steps))
