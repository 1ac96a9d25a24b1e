"""Debug: sparse segments against the twin, first mismatch per size."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mcmc_date_amd as M
import oracle as O
from mcmc_date_amd import synthetic as S

def problem(n_leaves, B):
    topo = S.random_topology(n_leaves, seed=9)
    n = topo.n_nodes - 2
    P, assoc = S.banded_precision(n, n, 3, 4)
    s0 = S.random_states(topo, B, seed=11)
    s0.time_birth_rate = np.full(B, 1.0); s0.time_death_rate = np.full(B, 0.8); s0.rate_variance = np.full(B, 0.3)
    D = np.array([O.distances(topo.parent, s0.heights[b], s0.rates[b], s0.time_height[b], s0.rate_mean[b]) for b in range(B)])
    mu = D.mean(axis=0)
    Pd = P.toarray()
    logdet = -float(np.linalg.slogdet(Pd)[1])
    return topo, P, Pd, assoc, mu, logdet, s0

for n_leaves, B, n_steps in [(7, 8, 700), (40, 16, 600), (200, 16, 200), (513, 64, 300), (1007, 8, 60)]:
    topo, P, Pd, assoc, mu, logdet, s0 = problem(n_leaves, B)
    tl = M.SparseLikelihood(M.Sparse(mu, assoc, logdet)).bind_tree(topo)
    pf = M.PriorFunction(1.0, "UncorrelatedGamma", [], [], [], topo)
    ps, _ = M.proposals(topo, [], calibrations_available=True)
    smp = M.Sampler(tl, pf, ps, B, seed=13)
    smp.set_state(s0)
    spec = O.PriorSpec(topo.parent, 1.0, "UncorrelatedGamma", [], [], [])
    twin = O.MhChains(O.MhModel(topo.parent, mu, Pd, logdet, spec, M.table_arrays(ps)), s0.time_birth_rate, s0.time_death_rate,
                      s0.time_height, s0.heights, s0.rate_mean, s0.rate_variance, s0.rates, seed=13)
    cyc = M.cycle_schedule(ps, 1, np.random.default_rng(0))
    sched = np.tile(cyc, (1, n_steps // cyc.shape[1] + 1))[:, :n_steps]
    ta, tk = smp.run_schedule(sched, trace=True)
    ra, rk = twin.run(sched, trace=True)
    print(n_leaves, B, n_steps, smp.last_path()[:40])
    fin = np.isfinite(ra)
    bad = np.argwhere(np.isfinite(ta) != fin)
    tol = 1e-8 + 1e-10 * np.abs(ra)
    with np.errstate(invalid="ignore"):
        bad2 = np.argwhere(fin & np.isfinite(ta) & (np.abs(ta - ra) > tol))
    print("  finite-pattern mismatches", len(bad), "value mismatches", len(bad2), "decision mismatches", int((tk != rk).sum()))
    for (st, b) in list(bad[:6]) + list(bad2[:6]):
        p = ps[sched[0, st]]
        print("   step", st, "chain", b, "prop", p.name, "kind", p.kind, "node", p.node, "device", ta[st, b], "twin", ra[st, b])
