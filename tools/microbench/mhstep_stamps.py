"""Phase stamps of k_mh_step (csrc/k_mh.hip), wave 0 of workgroup 0, summed per proposal kind: s_memtime ticks between
start | decision | state in LDS | proposal done | changed blocks known | node priors | birth-death | clock | stores landed
(shader cycles, about 1.85 GHz under this load: 21 500 cycles = the 11.6 us of a rate proposal's launch).
Build first:  make -C mcmc-date_amd/csrc stamp_mhstep   (writes tools/microbench/libmhstepstamp.so);
on the GPU box:  python tools/microbench/mhstep_stamps.py [n] [chains] [steps]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MCD_LIB_PATH"] = os.path.join(ROOT, "tools", "microbench", "libmhstepstamp.so")
sys.path.insert(0, ROOT)
import time
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
s0.time_birth_rate = np.full(B, 1.0)
s0.time_death_rate = np.full(B, 0.8)
s0.rate_variance = np.full(B, 0.3)
smp = M.Sampler(tl, pf, ps, B, seed=13)
smp.set_state(s0)
cyc = M.cycle_schedule(ps, 1, np.random.default_rng(0))
sched = np.tile(cyc, (1, steps // cyc.shape[1] + 1))[:, :steps]
smp.run_schedule(sched[:, :200])
torch.cuda.synchronize()
L = M._capi.lib()
acc0 = np.zeros(320, dtype=np.uint64); cnt0 = np.zeros(32, dtype=np.uint64)
L.mcd_mhstep_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
L.mcd_mhstep_debug_stamps(acc0.ctypes.data, cnt0.ctypes.data)
t0 = time.perf_counter()
smp.run_schedule(sched)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
acc = np.zeros(320, dtype=np.uint64); cnt = np.zeros(32, dtype=np.uint64)
L.mcd_mhstep_debug_stamps(acc.ctypes.data, cnt.ctypes.data)
acc = (acc - acc0).astype(np.float64).reshape(32, 10); cnt = (cnt - cnt0).astype(np.float64)
print("us per lock step %.2f  (n_nodes %d, chains %d)" % (1e6 * dt / steps, topo.n_nodes, B))
print("k_mh_step (<= 320 nodes), shader cycles per phase: decision | state->LDS | propose | changed? | nodes | bd | clock | stores | -")
print("k_mh_step_wg: first loads landed | decision | state loads landed | state in LDS | proposal's scalars | transform | distances, summands | columns | sums    total  count")
for k in range(32):
    if cnt[k] > 0:
        r = acc[k, :9] / cnt[k]
        print("kind %2d: " % k + " ".join("%7.1f" % v for v in r) + "   %7.1f  %d" % (r.sum(), cnt[k]))
