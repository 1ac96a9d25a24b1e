"""GPU parity of the lock-step Metropolis-Hastings-Green driver (first slice) against its CPU
twin oracle/mh_oracle.c.  Both sides draw the same counter-based random numbers, so chains can be compared step by
step: ln acceptance ratio of every step within 1e-8 + 1e-12 |ln posterior| + 1e-10 |ratio| (the device evaluates the
likelihood through the Cholesky factor and libm differs from ocml in the last ulp), identical accept/reject
decisions, identical counters, final states within 1e-9 relative.  Posterior level: node-age means of two
independent runs (device vs twin, different seeds) within 1 % on tests/12-leaves-variable-rate (north_star)."""
import numpy as np
import pytest

import mcmc_date_amd as M
import oracle as O

pytestmark = pytest.mark.gpu
FIX = ["06-leaves-constant-rate", "10-leaves-autocorrelated-rate", "12-leaves-variable-rate", "24-leaves-braces", "25-leaves-bastien"]


def setup(fx, model="UncorrelatedGamma", B=16, seed=7, first_chain=0):
    topo = M.Topology(fx["parent"])
    cal = [M.Calibration(f"c{i}", int(r[0]), r[2] if r[1] else None, r[3], r[5] if r[4] else None, r[6]) for i, r in enumerate(fx["cal"])]
    con = [M.Constraint(f"k{i}", int(r[0]), int(r[1]), r[2]) for i, r in enumerate(fx["con"])]
    br = [M.Brace(f"b{i}", [int(n) for n in fx["brace_nodes"][fx["brace_ptr"][i]:fx["brace_ptr"][i + 1]]], float(s))
          for i, s in enumerate(fx["brace_sd"])]
    ht = float(fx["prior_ht"])
    ps, missing = M.proposals(topo, br, calibrations_available=len(cal) > 0)
    lik = M.MvnLikelihood(M.Full(fx["mu"], fx["sigma_inv"], float(fx["logdet"]))).bind_tree(topo)
    pf = M.PriorFunction(ht, model, cal, con, br, topo)
    smp = M.Sampler(lik, pf, ps, B, seed, first_chain=first_chain)
    spec = O.PriorSpec(fx["parent"], ht, model, [(c.node, c.lower, c.lower_p, c.upper, c.upper_p) for c in cal],
                       [(k.young, k.old, k.p) for k in con], [(b.nodes, b.sd) for b in br])
    twin_model = O.MhModel(fx["parent"], fx["mu"], fx["sigma_inv"], float(fx["logdet"]), spec, M.table_arrays(ps))
    x0 = M.init_with(topo, fx["mean_lengths"])
    if cal:
        x0.time_height = ht
    s0 = M.StateBatch.from_states([x0] * B)
    smp.set_state(s0)
    twin = O.MhChains(twin_model, s0.time_birth_rate, s0.time_death_rate, s0.time_height, s0.heights, s0.rate_mean,
                      s0.rate_variance, s0.rates, seed=seed, chain0=first_chain)
    return topo, ps, smp, twin


def alpha_close(a, b, tol):
    """ln acceptance ratios agree: absolute `tol` plus 1e-10 relative (a contrary slide that lands next to a bound
    multiplies a rate by 1 / (tiny height difference): ratios of -1e5 and more, conditioned accordingly)."""
    return bool(np.all(np.abs(a - b) <= tol + 1e-10 * np.abs(b)))


def compare_states(smp, twin, rtol=1e-9, atol_post=1e-8):
    s = smp.state()
    for a, b in ((s.time_birth_rate, twin.birth), (s.time_death_rate, twin.death), (s.time_height, twin.tH), (s.heights, twin.H),
                 (s.rate_mean, twin.rMu), (s.rate_variance, twin.rVar), (s.rates, twin.R)):
        assert np.allclose(a, b, rtol=rtol, atol=0), np.max(np.abs(a - b) / np.maximum(1e-300, np.abs(b)))
    post = smp.posterior()
    assert np.allclose(post, twin.post, rtol=1e-13, atol=atol_post)


@pytest.mark.parametrize("name", FIX)
def test_lockstep_parity_with_cpu_twin(gpu, golden, name):
    topo, ps, smp, twin = setup(golden[name], B=16, seed=11)
    rng = np.random.default_rng(5)
    # every chain its own tuning parameters, spread over two orders of magnitude
    tune = np.exp(rng.uniform(np.log(0.1), np.log(10.0), (16, len(ps))))
    smp.set_tuning(tune)
    twin.tune[:] = tune
    sched = M.cycle_schedule(ps, 4, rng)
    # far from the mode (the initial state) ln likelihoods reach 1e10: the tolerance scales with that magnitude
    tol = 1e-8 + 1e-12 * np.abs(smp.posterior()[:, :2]).max()
    ta, tk = smp.run_schedule(sched, accumulate=True, trace=True)
    ra, rk = twin.run(sched, accumulate=True, trace=True)
    fin = np.isfinite(ra)
    assert np.array_equal(np.isfinite(ta), fin)
    assert alpha_close(ta[fin], ra[fin], tol)
    assert np.array_equal(tk, rk)
    assert 0.05 < tk.mean() < 0.95                      # both outcomes exercised
    t, acc, tried = smp.tuning()
    assert np.array_equal(acc, twin.acc) and np.array_equal(tried, twin.tried) and np.array_equal(t, tune)
    compare_states(smp, twin)
    s, q, n = smp.age_sums()
    assert n == 4 and np.allclose(s, twin.age_sum, rtol=1e-9) and np.allclose(q, twin.age_sq, rtol=1e-9)
    # auto tuning
    smp.autotune()
    twin.autotune()
    t, acc, tried = smp.tuning()
    assert np.allclose(t, twin.tune, rtol=1e-14) and not acc.any() and not tried.any()
    # a second stretch continues the same random streams
    sched = M.cycle_schedule(ps, 2, rng)
    ta, tk = smp.run_schedule(sched, trace=True)
    ra, rk = twin.run(sched, trace=True)
    assert np.array_equal(tk, rk)
    compare_states(smp, twin)


@pytest.mark.parametrize("model", ["UncorrelatedLogNormal", "UncorrelatedWhiteNoise", "AutocorrelatedLogNormal"])
def test_lockstep_parity_other_clock_models(gpu, golden, model):
    topo, ps, smp, twin = setup(golden["12-leaves-variable-rate"], model=model, B=8, seed=3)
    sched = M.cycle_schedule(ps, 3, np.random.default_rng(1))
    tol = 1e-8 + 1e-12 * np.abs(smp.posterior()[:, :2]).max()
    ta, tk = smp.run_schedule(sched, trace=True)
    ra, rk = twin.run(sched, trace=True)
    fin = np.isfinite(ra)
    assert np.array_equal(np.isfinite(ta), fin) and alpha_close(ta[fin], ra[fin], tol)
    assert np.array_equal(tk, rk)
    compare_states(smp, twin)


def test_chain_kernel_equals_per_phase_kernels(gpu, golden, knobs):
    """Trees of at most 64 nodes run the whole schedule in one launch (k_mh_chain.hip); larger trees use two
    launches per step (k_mh.hip + k_tree_logpdf.hip).  Same arithmetic in the same order: bit-identical chains."""
    fx = golden["25-leaves-bastien"]
    topo, ps, fused, _ = setup(fx, B=12, seed=5)
    sched = M.cycle_schedule(ps, 3, np.random.default_rng(8))
    a1, k1 = fused.run_schedule(sched, accumulate=True, trace=True)
    assert fused.last_path().startswith("whole schedule in one launch, factor resident in LDS")
    knobs.setenv("MCD_MH_PER_PHASE", "1")             # (read when the handle is made and at every run)
    _, _, phased, _ = setup(fx, B=12, seed=5)
    a2, k2 = phased.run_schedule(sched, accumulate=True, trace=True)
    assert phased.last_path().startswith("two launches per lock step")
    knobs.delenv("MCD_MH_PER_PHASE")
    assert np.array_equal(a1, a2, equal_nan=True) and np.array_equal(k1, k2)
    s1, s2 = fused.state(), phased.state()
    for f in ("heights", "rates", "time_height", "rate_mean", "rate_variance", "time_birth_rate", "time_death_rate"):
        assert np.array_equal(getattr(s1, f), getattr(s2, f))
    assert np.array_equal(fused.posterior(), phased.posterior())
    assert all(np.array_equal(x, y) for x, y in zip(fused.tuning(), phased.tuning()))
    assert all(np.array_equal(x, y) for x, y in zip(fused.age_sums()[:2], phased.age_sums()[:2]))


@pytest.mark.parametrize("dataset", ["12-leaves-variable-rate", "25-leaves-bastien"])
def test_chain_kernel_with_a_likelihood_wave(gpu, golden, dataset, knobs):
    """Trees of at most 64 nodes, fewer than 1024 chains: a second wave per chain evaluates the ln likelihood of the proposed state while
    the chain's wave evaluates the ln prior (k_mh_chain.hip, LW).  The same instructions on the same numbers as the one-wave kernel
    (MCD_MH_CHAIN_LW=0): bit-identical traces, states, posteriors, counters and age sums -- with calibrations, constraints, braces."""
    fx = golden[dataset]
    runs = []
    for lw in ("1", "0"):
        knobs.setenv("MCD_MH_CHAIN_LW", lw)
        topo, ps, smp, _ = setup(fx, B=37, seed=5)
        sched = M.cycle_schedule(ps, 3, np.random.default_rng(8))
        a, k = smp.run_schedule(sched, accumulate=True, trace=True)
        assert smp.last_path().startswith("whole schedule in one launch, factor resident in LDS")
        runs.append((a, k, smp.state(), smp.posterior(), smp.tuning(), smp.age_sums()[:2]))
    (a1, k1, s1, p1, t1, g1), (a2, k2, s2, p2, t2, g2) = runs
    assert np.array_equal(a1, a2, equal_nan=True) and np.array_equal(k1, k2) and 0.02 < k1.mean() < 0.98
    for f in ("heights", "rates", "time_height", "rate_mean", "rate_variance", "time_birth_rate", "time_death_rate"):
        assert np.array_equal(getattr(s1, f), getattr(s2, f)), f
    assert np.array_equal(p1, p2) and all(np.array_equal(x, y) for x, y in zip(t1, t2)) and all(np.array_equal(x, y) for x, y in zip(g1, g2))


@pytest.mark.parametrize("n_leaves,B", [(12, 10), (70, 6), (129, 512), (129, 700), (200, 40), (513, 96)])
def test_prior_beside_the_likelihood_gives_the_same_chains(gpu, n_leaves, B, knobs):
    """Two-launch path: by default the likelihood launch carries the ln prior of the proposed states as workgroups of a second
    role (k_tree_logpdf.hip PRIOR variant, mh_prior_role.hpp); MCD_MH_PRIOR=0 evaluates it inside k_mh_step as before.  The same
    wave-level functions on the same numbers: bit-identical acceptance ratios, decisions, states, posteriors, age sums -- with
    calibrations, constraints and every clock model's blocks in play (257 nodes: 512 chains = two compute waves per
    workgroup, 700 = four).  Above N = 256 (399 and 1025 nodes here) the likelihood is the row-split kernel and the prior
    stays inside the step: both settings are then the same launches."""
    from mcmc_date_amd import synthetic as S

    knobs.setenv("MCD_MH_PER_PHASE", "1")              # (the 23-node tree would otherwise run the whole-schedule kernel)
    topo = S.random_topology(n_leaves, seed=21)
    n = topo.n_nodes - 2
    mu, sigma = S.random_spd_problem(n, seed=21)
    s0 = S.random_states(topo, B, seed=22)
    s0.time_birth_rate = np.full(B, 1.0); s0.time_death_rate = np.full(B, 0.8); s0.rate_variance = np.full(B, 0.3)
    cal = [M.Calibration("root", 0, 0.9, 0.025, 1.1, 0.025), M.Calibration("n", 5, 0.2, 0.025, None, 0.0)]
    con = [M.Constraint("k", 7, 3, 0.025)]
    ps, _ = M.proposals(topo, [], calibrations_available=True)
    sched = M.cycle_schedule(ps, 1, np.random.default_rng(0))[:, :300]
    for model in ("UncorrelatedGamma", "AutocorrelatedLogNormal"):
        runs = []
        for inside in (False, True):
            if inside:
                knobs.setenv("MCD_MH_PRIOR", "0")
            else:
                knobs.delenv("MCD_MH_PRIOR", raising=False)
            lik = M.MvnLikelihood.from_covariance(mu, sigma).bind_tree(topo)
            smp = M.Sampler(lik, M.PriorFunction(1.0, model, cal, con, [], topo), ps, B, seed=13)
            smp.set_state(s0)
            a, k = smp.run_schedule(sched, accumulate=True, trace=True)
            runs.append((a, k, smp.state(), smp.posterior(), smp.age_sums()[:2]))
        knobs.delenv("MCD_MH_PRIOR", raising=False)
        (a1, k1, s1, p1, g1), (a2, k2, s2, p2, g2) = runs
        assert np.array_equal(a1, a2, equal_nan=True) and np.array_equal(k1, k2) and 0.02 < k1.mean() < 0.98
        for f in ("heights", "rates", "time_height", "rate_mean", "rate_variance", "time_birth_rate", "time_death_rate"):
            assert np.array_equal(getattr(s1, f), getattr(s2, f)), (model, f)
        assert np.array_equal(p1, p2) and all(np.array_equal(x, y) for x, y in zip(g1, g2))


@pytest.mark.parametrize("n_leaves,B", [(12, 10), (129, 64), (200, 40), (513, 96)])
def test_workgroup_per_chain_step_gives_the_same_chains(gpu, n_leaves, B, knobs):
    """Trees of more than 320 nodes take the step kernel with a workgroup of four waves per chain (k_mh_step_wg: threads = nodes
    for the copies, worker waves for the per-node summands of the ln prior, added up in the one-wave order); MCD_MH_STEP_WG=1 / 0
    force it / the one-wave kernel for any tree.  Bit-identical chains: traces, states, posteriors, age sums."""
    from mcmc_date_amd import synthetic as S

    knobs.setenv("MCD_MH_PER_PHASE", "1")
    knobs.setenv("MCD_MH_INCREMENTAL", "0")            # (the incremental likelihood of the large-tree path has its own test below)
    topo = S.random_topology(n_leaves, seed=41)
    n = topo.n_nodes - 2
    mu, sigma = S.random_spd_problem(n, seed=41)
    s0 = S.random_states(topo, B, seed=42)
    s0.time_birth_rate = np.full(B, 1.0); s0.time_death_rate = np.full(B, 0.8); s0.rate_variance = np.full(B, 0.3)
    cal = [M.Calibration("root", 0, 0.9, 0.025, 1.1, 0.025), M.Calibration("n", 5, 0.2, 0.025, None, 0.0)]
    con = [M.Constraint("k", 7, 3, 0.025)]
    ps, _ = M.proposals(topo, [], calibrations_available=True)
    sched = M.cycle_schedule(ps, 1, np.random.default_rng(0))[:, :300]
    for model, inside in (("UncorrelatedGamma", True), ("UncorrelatedWhiteNoise", True), ("UncorrelatedGamma", False)):
        if inside:
            knobs.setenv("MCD_MH_PRIOR", "0")          # the prior inside the step: the part the two kernels do differently
        else:
            knobs.delenv("MCD_MH_PRIOR", raising=False)   # beside the likelihood (N <= 256): the kernels only leave the flags
        runs = []
        for wg in ("1", "0"):
            knobs.setenv("MCD_MH_STEP_WG", wg)
            lik = M.MvnLikelihood.from_covariance(mu, sigma).bind_tree(topo)
            smp = M.Sampler(lik, M.PriorFunction(1.0, model, cal, con, [], topo), ps, B, seed=13)
            smp.set_state(s0)
            a, k = smp.run_schedule(sched, accumulate=True, trace=True)
            runs.append((a, k, smp.state(), smp.posterior(), smp.age_sums()[:2]))
        (a1, k1, s1, p1, g1), (a2, k2, s2, p2, g2) = runs
        assert np.array_equal(a1, a2, equal_nan=True) and np.array_equal(k1, k2) and 0.02 < k1.mean() < 0.98
        for f in ("heights", "rates", "time_height", "rate_mean", "rate_variance", "time_birth_rate", "time_death_rate"):
            assert np.array_equal(getattr(s1, f), getattr(s2, f)), (model, f)
        assert np.array_equal(p1, p2) and all(np.array_equal(x, y) for x, y in zip(g1, g2))


@pytest.mark.parametrize("n_leaves,B", [(33, 9), (40, 7), (70, 64), (129, 512), (129, 33), (100, 16), (129, 777)])
@pytest.mark.parametrize("incremental", ["1", "0"])
def test_streaming_chain_kernel_equals_two_launch_path(gpu, n_leaves, B, incremental, knobs):
    """Trees of 65 .. 258 nodes at up to 1024 chains (777: two rounds of workgroups) run the whole schedule in one launch, two
    chains per workgroup, the factor streamed through the sweep's LDS ring (k_mh_chain_big.hip; round 3 also built it for six and
    eight 64-row blocks per lane -- up to 514 nodes --, which spilled and which the segment kernel superseded: removed in round 4,
    those trees are covered by test_incremental_likelihood_on_large_trees and test_large_tree_uses_the_per_phase_path).  The same proposal, prior and sweep
    code on the same numbers as the two-launch path (MCD_MH_PER_PHASE=1) -- odd batches (a chain wave without a chain), two clock
    models, calibrations and a constraint, runs continued by the other path.
    MCD_MH_INCREMENTAL=0 (every proposal through the full sweep): bit-identical traces, states, posteriors, tuning counters, age
    sums.  Default (round 3: proposals that move a few distances are evaluated by columns of L^-1 on the kept z = L^-1 (d - mu)):
    the ln likelihood agrees with the full evaluation to rounding, so the traced ln acceptance ratios agree within the twin's
    tolerance, and the accept / reject decisions, states, ln priors, ln Jacobians, counters are still the same bits."""
    from mcmc_date_amd import synthetic as S

    knobs.setenv("MCD_MH_INCREMENTAL", incremental)
    exact = incremental == "0"
    topo = S.random_topology(n_leaves, seed=51)
    n = topo.n_nodes - 2
    mu, sigma = S.random_spd_problem(n, seed=51)
    s0 = S.random_states(topo, B, seed=52)
    s0.time_birth_rate = np.full(B, 1.0); s0.time_death_rate = np.full(B, 0.8); s0.rate_variance = np.full(B, 0.3)
    cal = [M.Calibration("root", 0, 0.9, 0.025, 1.1, 0.025), M.Calibration("n", 5, 0.2, 0.025, None, 0.0)]
    con = [M.Constraint("k", 7, 3, 0.025)]
    braces = []
    if B == 33:                                              # braced nodes (Brace.hs): a pair of internal nodes, one from each side of the root
        par = np.asarray(topo.parent)
        size = np.ones(topo.n_nodes, int)
        for v in range(topo.n_nodes - 1, 0, -1):
            size[par[v]] += size[v]
        rr = 1 + size[1]
        braces = [M.Brace("b0", [[v for v in range(2, rr) if size[v] > 1][0], [v for v in range(rr + 1, topo.n_nodes) if size[v] > 1][0]], 0.01)]
    ps, _ = M.proposals(topo, braces, calibrations_available=True)
    sched = M.cycle_schedule(ps, 1, np.random.default_rng(0))
    if braces:                                               # (the brace proposals of the cycle among the 260 steps)
        kinds = M.table_arrays(ps)["kind"][sched[0]]
        pick = np.sort(np.unique(np.concatenate([np.flatnonzero(np.isin(kinds, [14, 15]))[:20], np.arange(240)])))
        sched = sched[:, pick[:260]]
    else:
        sched = sched[:, :260]
    for model in ("UncorrelatedGamma", "AutocorrelatedLogNormal"):
        runs = []
        for phased in (False, True):
            if phased:
                knobs.setenv("MCD_MH_PER_PHASE", "1")
            else:
                knobs.delenv("MCD_MH_PER_PHASE", raising=False)
            lik = M.MvnLikelihood.from_covariance(mu, sigma).bind_tree(topo)
            lik.mvn.set_form("sweep")                        # (the two-launch path would take the row split at 240 < N <= 256, <= 128 chains)
            smp = M.Sampler(lik, M.PriorFunction(1.0, model, cal, con, braces, topo), ps, B, seed=13)
            smp.set_state(s0)
            tol = 1e-8 + 1e-12 * np.abs(smp.posterior()[:, :2]).max()
            a, k = smp.run_schedule(sched[:, :200], accumulate=True, trace=True)
            if n_leaves > 32 and not phased:
                assert smp.last_path().startswith("whole schedule in one launch, two chains per workgroup")
            # the last 60 steps by the OTHER path: what one path leaves behind, the other continues from
            if phased:
                knobs.delenv("MCD_MH_PER_PHASE", raising=False)
            else:
                knobs.setenv("MCD_MH_PER_PHASE", "1")
            a2, k2 = smp.run_schedule(sched[:, 200:], accumulate=True, trace=True)
            runs.append((np.concatenate([a, a2]), np.concatenate([k, k2]), smp.state(), smp.posterior(), smp.tuning(), smp.age_sums()[:2]))
        knobs.delenv("MCD_MH_PER_PHASE", raising=False)
        (a1, k1, s1, p1, t1, g1), (a2, k2, s2, p2, t2, g2) = runs
        assert np.array_equal(k1, k2) and 0.02 < k1.mean() < 0.98
        fin = np.isfinite(a2)
        assert np.array_equal(np.isfinite(a1), fin)
        if exact:
            assert np.array_equal(a1, a2, equal_nan=True) and np.array_equal(p1, p2)
        else:
            assert alpha_close(a1[fin], a2[fin], tol), np.max(np.abs(a1[fin] - a2[fin]))
            assert np.array_equal(p1[:, [0, 2]], p2[:, [0, 2]]) and np.allclose(p1[:, 1], p2[:, 1], rtol=1e-12, atol=tol)
        for f in ("heights", "rates", "time_height", "rate_mean", "rate_variance", "time_birth_rate", "time_death_rate"):
            assert np.array_equal(getattr(s1, f), getattr(s2, f)), (model, f)
        assert all(np.array_equal(x, y) for x, y in zip(t1, t2))
        assert all(np.allclose(x, y, rtol=1e-13, atol=0) for x, y in zip(g1, g2))      # (sums formed per run, then added: not bitwise)


@pytest.mark.parametrize("n_leaves,B", [(200, 40), (513, 64), (513, 5), (300, 33), (300, 1100), (200, 2100)])
def test_incremental_likelihood_on_large_trees(gpu, n_leaves, B, knobs):
    """Trees of more than 320 nodes at a sampler's batch (399, 599 and 1025 nodes here; an odd batch leaves a workgroup of the segment
    kernel with one chain; 1100 chains: beyond one launch of the row-split kernel, its z products are taken chunk by chunk): the workgroup-per-chain step kernel with the likelihood launch only for the proposals that move many
    distances (k_mh_inc.hip; the others: columns of L^-1 on a z kept in global memory, refreshed by a full product every 256 steps)
    and the runs between two dense proposals in one launch each (k_mh_segment.hip; default from 259 nodes), against the same path with the
    full evaluation at every step (MCD_MH_INCREMENTAL=0): 1 500 lock steps, identical accept / reject decisions, states, ln priors
    and ln Jacobians bit for bit, ln acceptance ratios and ln likelihoods within the twin's tolerance."""
    from mcmc_date_amd import synthetic as S

    knobs.setenv("MCD_MH_PER_PHASE", "1")              # (399 nodes would otherwise take the streaming chain kernel)
    topo = S.random_topology(n_leaves, seed=61)
    n = topo.n_nodes - 2
    mu, sigma = S.random_spd_problem(n, seed=61)
    s0 = S.random_states(topo, B, seed=62)
    s0.time_birth_rate = np.full(B, 1.0); s0.time_death_rate = np.full(B, 0.8); s0.rate_variance = np.full(B, 0.3)
    cal = [M.Calibration("root", 0, 0.9, 0.025, 1.1, 0.025), M.Calibration("n", 5, 0.2, 0.025, None, 0.0)]
    braces = []
    if B == 33:                                              # braced nodes (Brace.hs): two pairs of internal nodes, one from each side of the root
        par = np.asarray(topo.parent)
        size = np.ones(topo.n_nodes, int)
        for v in range(topo.n_nodes - 1, 0, -1):
            size[par[v]] += size[v]
        rr = 1 + size[1]
        left = [v for v in range(2, rr) if size[v] > 1][:2]
        right = [v for v in range(rr + 1, topo.n_nodes) if size[v] > 1][:2]
        braces = [M.Brace("b0", [left[0], right[0]], 0.01), M.Brace("b1", [left[1], right[1]], 0.02)]
    ps, _ = M.proposals(topo, braces, calibrations_available=True)
    sched = M.cycle_schedule(ps, 1, np.random.default_rng(0))
    if braces:                                               # (every brace proposal of the cycle among the 1 500 steps)
        kinds = M.table_arrays(ps)["kind"][sched[0]]
        first = np.concatenate([np.flatnonzero(np.isin(kinds, [14, 15]))[:40], np.arange(1460)])
        sched = sched[:, np.sort(np.unique(first))[:1500]]
    else:
        sched = sched[:, :1500]
    runs = []
    for inc, seg in (("1", "1"), ("1", "0"), ("0", "0")):
        knobs.setenv("MCD_MH_INCREMENTAL", inc)
        knobs.setenv("MCD_MH_SEGMENTS", seg)
        lik = M.MvnLikelihood.from_covariance(mu, sigma).bind_tree(topo)
        smp = M.Sampler(lik, M.PriorFunction(1.0, "UncorrelatedGamma", cal, [], braces, topo), ps, B, seed=13)
        smp.set_state(s0)
        if B == 33:                                          # heated chains (MC3): posterior^beta in every path's acceptance ratio
            smp.set_temperatures(np.linspace(1.0, 0.55, B))
        tol = 1e-8 + 1e-12 * np.abs(smp.posterior()[:, :2]).max()
        a, k = smp.run_schedule(sched[:, :700], accumulate=True, trace=True)
        a2, k2 = smp.run_schedule(sched[:, 700:], accumulate=True, trace=True)       # a second call starts from a fresh full product
        # from 259 nodes: the runs between two dense proposals in one launch each (k_mh_segment.hip) unless switched off
        want = "segments" if (inc == "1" and seg == "1" and topo.n_nodes > 258) else "only for proposals that move many distances" if inc == "1" else "plain-vector likelihood"
        if want == "segments" or B <= 1024:                  # (beyond 1024 chains the paths without segments follow the likelihood launch's form)
            assert want in smp.last_path(), smp.last_path()
        runs.append((np.concatenate([a, a2]), np.concatenate([k, k2]), smp.state(), smp.posterior(), smp.tuning(), smp.age_sums()[:2]))
    (a2, k2, s2, p2, t2, g2) = runs[-1]
    for (a1, k1, s1, p1, t1, g1) in runs[:-1]:
        assert np.array_equal(k1, k2) and 0.02 < k1.mean() < 0.98
        fin = np.isfinite(a2)
        assert np.array_equal(np.isfinite(a1), fin) and alpha_close(a1[fin], a2[fin], tol), np.max(np.abs(a1[fin] - a2[fin]))
        for f in ("heights", "rates", "time_height", "rate_mean", "rate_variance", "time_birth_rate", "time_death_rate"):
            assert np.array_equal(getattr(s1, f), getattr(s2, f)), f
        assert np.array_equal(p1[:, [0, 2]], p2[:, [0, 2]]) and np.allclose(p1[:, 1], p2[:, 1], rtol=1e-12, atol=tol)
        assert all(np.array_equal(x, y) for x, y in zip(t1, t2)) and all(np.array_equal(x, y) for x, y in zip(g1, g2))


def test_streaming_chain_kernel_over_many_steps_against_the_twin(gpu):
    """The incremental evaluation over a long stretch: 257 nodes x 16 chains, three iterations of the cycle (12 099 lock steps: z is
    updated by columns of L^-1 between the sweeps of dense proposals and the refresh every 256 steps) against the CPU twin, which
    evaluates every proposal in full -- identical decisions at every step, ln acceptance ratios within the twin's tolerance, final
    states within 1e-9, the final ln likelihood within 1e-10 relative."""
    from mcmc_date_amd import synthetic as S

    topo = S.random_topology(129, seed=7)
    n = topo.n_nodes - 2
    mu, sigma = S.random_spd_problem(n, seed=7)
    sigma_inv = np.linalg.inv(sigma)
    logdet = float(np.linalg.slogdet(sigma)[1])
    B = 16
    s0 = S.random_states(topo, B, seed=8)
    s0.time_birth_rate = np.full(B, 1.0); s0.time_death_rate = np.full(B, 0.8); s0.rate_variance = np.full(B, 0.3)
    ps, _ = M.proposals(topo, [], calibrations_available=True)
    lik = M.MvnLikelihood(M.Full(mu, sigma_inv, logdet)).bind_tree(topo)
    lik.mvn.set_form("sweep")
    pf = M.PriorFunction(1.0, "UncorrelatedGamma", [], [], [], topo)
    smp = M.Sampler(lik, pf, ps, B, seed=13)
    smp.set_state(s0)
    spec = O.PriorSpec(topo.parent, 1.0, "UncorrelatedGamma", [], [], [])
    twin = O.MhChains(O.MhModel(topo.parent, mu, sigma_inv, logdet, spec, M.table_arrays(ps)), s0.time_birth_rate, s0.time_death_rate,
                      s0.time_height, s0.heights, s0.rate_mean, s0.rate_variance, s0.rates, seed=13)
    sched = M.cycle_schedule(ps, 3, np.random.default_rng(0))
    tol = 1e-8 + 1e-12 * np.abs(smp.posterior()[:, :2]).max()
    ta, tk = smp.run_schedule(sched, trace=True)
    assert smp.last_path().startswith("whole schedule in one launch, two chains per workgroup")
    ra, rk = twin.run(sched, trace=True)
    fin = np.isfinite(ra)
    assert np.array_equal(np.isfinite(ta), fin) and alpha_close(ta[fin], ra[fin], tol)
    assert np.array_equal(tk, rk) and 0.02 < tk.mean() < 0.98
    compare_states(smp, twin, atol_post=tol)
    assert np.allclose(smp.posterior()[:, 1], twin.post[:, 1], rtol=1e-10, atol=0)


@pytest.mark.parametrize("n_leaves,B,n_steps", [(70, 6, 400), (128, 64, 150), (70, 2100, 24), (140, 33, 150), (150, 512, 60), (160, 64, 100),
                                                (513, 512, 24), (513, 32, 600), (300, 16, 600), (513, 8, 3000)])
def test_large_tree_uses_the_per_phase_path(gpu, n_leaves, B, n_steps):
    """Synthetic trees beyond 64 nodes (70 leaves: 139 nodes, N = 137, three row blocks; 128 leaves: 255 nodes, N = 253,
    the size of BASELINE.json's config 3, with 64 chains): lanes stride over the nodes and the likelihood runs through
    the streaming kernel; parity with the CPU twin as for the small trees.  With 2100 chains the
    likelihood launch takes the multiply form on the matrix cores (k_wide.hip).  140 / 150 / 160 leaves = 279 / 299 / 319 nodes:
    the window between the streaming chain kernel (N <= 256, i.e. <= 258 nodes) and the workgroup-per-chain step kernel (more
    than 320 nodes) -- one-wave k_mh_step with the prior inside + the row-split likelihood (round-2 review: no test reached it).
    513 leaves x 512 chains = config 5's share of one GPU (1025 nodes), step by step against the twin; 600 lock steps at 1025 and at 599
    nodes: several segments (k_mh_segment.hip), dense proposals between them, two recomputations of z."""
    from mcmc_date_amd import synthetic as S

    topo = S.random_topology(n_leaves, seed=3)
    n = topo.n_nodes - 2
    mu, sigma = S.random_spd_problem(n, seed=3)
    sigma_inv = np.linalg.inv(sigma)
    logdet = float(np.linalg.slogdet(sigma)[1])
    s0 = S.random_states(topo, B, seed=4)
    s0.time_birth_rate = np.full(B, 1.0); s0.time_death_rate = np.full(B, 0.8); s0.rate_variance = np.full(B, 0.3)
    ps, _ = M.proposals(topo, [], calibrations_available=True)
    lik = M.MvnLikelihood(M.Full(mu, sigma_inv, logdet)).bind_tree(topo)
    pf = M.PriorFunction(1.0, "UncorrelatedGamma", [], [], [], topo)
    smp = M.Sampler(lik, pf, ps, B, seed=13)
    smp.set_state(s0)
    spec = O.PriorSpec(topo.parent, 1.0, "UncorrelatedGamma", [], [], [])
    twin = O.MhChains(O.MhModel(topo.parent, mu, sigma_inv, logdet, spec, M.table_arrays(ps)), s0.time_birth_rate, s0.time_death_rate,
                      s0.time_height, s0.heights, s0.rate_mean, s0.rate_variance, s0.rates, seed=13)
    sched = M.cycle_schedule(ps, 1, np.random.default_rng(0))[:, :n_steps]
    tol = 1e-8 + 1e-12 * np.abs(smp.posterior()[:, :2]).max()
    ta, tk = smp.run_schedule(sched, trace=True)
    ra, rk = twin.run(sched, trace=True)
    fin = np.isfinite(ra)
    assert np.array_equal(np.isfinite(ta), fin) and alpha_close(ta[fin], ra[fin], tol)
    assert np.array_equal(tk, rk) and 0.02 < tk.mean() < 0.98
    compare_states(smp, twin, atol_post=tol)


def test_chain_streams_do_not_depend_on_the_shard(gpu, golden):
    """Chains 8..15 of a 16-chain run equal an 8-chain shard created with first_chain = 8 (what a second GPU runs)."""
    fx = golden["12-leaves-variable-rate"]
    topo, ps, full, _ = setup(fx, B=16, seed=21)
    _, _, shard, _ = setup(fx, B=8, seed=21, first_chain=8)
    sched = M.cycle_schedule(ps, 3, np.random.default_rng(2))
    _, k_full = full.run_schedule(sched, trace=True)
    _, k_shard = shard.run_schedule(sched, trace=True)
    assert np.array_equal(k_full[:, 8:], k_shard)
    a, b = full.state(), shard.state()
    assert np.array_equal(a.heights[8:], b.heights) and np.array_equal(a.rates[8:], b.rates) and np.array_equal(a.time_height[8:], b.time_height)


def test_rejects_invalid_states_and_tables(gpu, golden):
    fx = golden["12-leaves-variable-rate"]
    topo, ps, smp, twin = setup(fx, B=4, seed=1)
    # a chain that starts outside the support never moves: every proposal is rejected (NaN / -inf ratios)
    s = smp.state()
    s.time_height[2] = -1.0
    smp.set_state(s)
    _, tk = smp.run_schedule(M.cycle_schedule(ps, 1, np.random.default_rng(0)), trace=True)
    after = smp.state()
    assert np.isneginf(smp.posterior()[2, 0])
    assert tk[:, 0].any() and not tk[:, 2].any()
    assert after.time_height[2] == -1.0 and np.array_equal(after.heights[2], s.heights[2]) and np.array_equal(after.rates[2], s.rates[2])
    # structural faults
    lik, pf = smp._keep
    leaf = int(np.nonzero(topo.leaves)[0][0])
    with pytest.raises(M.McdError, match="leaf"):
        M.Sampler(lik, pf, [M.Proposal("bad", M.sampler.SLIDE_NODE, leaf, 0.01)], 4, 0)
    with pytest.raises(M.McdError):
        M.Sampler(lik, pf, [M.Proposal("bad", 99, 1, 0.01)], 4, 0)
    with pytest.raises(M.McdError):
        M.Sampler(lik, pf, [M.Proposal("bad", M.sampler.SCALE_SCALAR, 7, 10.0)], 4, 0)
    with pytest.raises(M.McdError):
        smp.run_schedule(np.array([[len(ps)]], np.int32))
    fresh = M.Sampler(lik, pf, ps, 4, 0)
    with pytest.raises(M.McdError, match="set_state"):
        fresh.run(1)
    # a prior built for another topology with the same number of nodes
    from mcmc_date_amd import synthetic as S
    other = S.random_topology(12, seed=99)
    assert other.n_nodes == topo.n_nodes and not np.array_equal(other.parent, topo.parent)
    pf_other = M.PriorFunction(float(fx["prior_ht"]), "UncorrelatedGamma", [], [], [], other)
    with pytest.raises(M.McdError, match="different topologies"):
        M.Sampler(lik, pf_other, ps, 4, 0)


def test_posterior_node_ages_within_one_percent(gpu, golden):
    """north_star: posterior node-age means within 1 % of the CPU path on tests/12-leaves-variable-rate.  64 chains
    (BASELINE.json configs[1]), the reference's burn-in schedule and 8000 iterations (app/Definitions.hs:420-441);
    device run and CPU twin (32 chains, the slow side; it runs concurrently in a thread) use DIFFERENT seeds, i.e. the
    comparison is between two independent samples.  Measured: the largest relative difference over the inner nodes is 0.12 %."""
    import threading

    fx = golden["12-leaves-variable-rate"]
    B = 64
    topo, ps, smp, _ = setup(fx, B=B, seed=1001)
    _, _, _, twin = setup(fx, B=32, seed=2002)              # the CPU twin runs half as many chains (it is the slow side)

    def cpu_side():
        rng = np.random.default_rng(9)
        for period in M.sampler.BURN_IN_FAST + M.sampler.BURN_IN_SLOW:
            twin.run(M.cycle_schedule(ps, period, rng))
            twin.autotune()
        twin.run(M.cycle_schedule(ps, M.sampler.ITERATIONS, rng), accumulate=True)

    th = threading.Thread(target=cpu_side)                   # ctypes releases the GIL: both samplers run at the same time
    th.start()
    smp.burn_in()
    smp.run(M.sampler.ITERATIONS, accumulate=True)
    th.join()
    mean_gpu, var_gpu, sem_gpu = smp.node_age_summary()
    mean_cpu = twin.age_sum.mean(axis=0) / twin.n_samples
    inner = ~topo.leaves
    rel = np.abs(mean_gpu[inner] - mean_cpu[inner]) / mean_cpu[inner]
    print("max relative difference of node-age means: %.4f" % rel.max())
    assert rel.max() <= 0.01, rel
    # acceptance rates sit near the auto tuner's targets after burn-in
    t, acc, tried = smp.tuning()
    rate = acc.sum(axis=0) / np.maximum(1, tried.sum(axis=0))
    dims = np.array([p.dim for p in ps])
    assert np.all(np.abs(rate[dims == 1] - 0.44) < 0.1)


def test_monitor_files_and_age_summary(gpu, golden, tmp_path):
    """Row f4, reporting: the four monitor files of app/Definitions.hs:288-417 (period 2) for one chain and the node-age
    summary of the reference's analysis script, fed by the device sampler."""
    from mcmc_date_amd import monitor as MO

    fx = golden["24-leaves-braces"]
    topo, ps, smp, _ = setup(fx, B=8, seed=4)
    lik, pf = smp._keep
    cal = [M.Calibration(f"c{i}", int(r[0]), r[2] if r[1] else None, r[3], r[5] if r[4] else None, r[6]) for i, r in enumerate(fx["cal"])]
    con = [M.Constraint(f"k{i}", int(r[0]), int(r[1]), r[2]) for i, r in enumerate(fx["con"])]
    br = [M.Brace(f"b{i}", [int(n) for n in fx["brace_nodes"][fx["brace_ptr"][i]:fx["brace_ptr"][i + 1]]], float(s)) for i, s in enumerate(fx["brace_sd"])]
    smp.run(20)
    tr = MO.collect(smp, 41, accumulate=True)                 # 20 samples (period 2); the odd last iteration is run, not sampled
    assert list(tr.iteration) == list(range(22, 62, 2)) and smp.iterations_done == 61 and tr.heights.shape == (20, 8, topo.n_nodes)
    s_sum, _, n_acc = smp.age_sums()
    assert n_acc == 41
    files = MO.write_monitor_files(str(tmp_path / "run"), tr, 3, topo, cal, con, br, prior=pf)
    assert [f.rsplit(".", 2)[1] for f in files] == ["params", "timetree", "ratetree", "prior"]
    rows = [l.rstrip("\n").split("\t") for l in open(files[0])]
    assert rows[0][:6] == ["Iteration", "TimeBirthRate", "TimeDeathRate", "TimeHeight", "RateMean", "RateVariance"]
    assert len(rows) == 21 and len(rows[0]) == 6 + len(cal) + len(con) + len(br) and rows[1][0] == "22"
    assert float(rows[5][3]) == tr.time_height[4, 3]
    k = 7
    trees = [l.rstrip("\n").split("\t") for l in open(files[1])]
    from mcmc_date_amd.tree import parse_newick
    topo2, ln2 = parse_newick(trees[1 + k][1])[:2]
    assert np.array_equal(topo2.parent, topo.parent)
    assert np.allclose(ln2[1:], (M.height_tree_to_length_tree(topo, tr.heights[k, 3]) * tr.time_height[k, 3])[1:], rtol=1e-15)
    prior_rows = [l.split("\t") for l in open(files[3])]
    lp, comp = pf.logprior(tr.states(k).slice(3, 4), want_components=True)
    assert np.allclose([float(x) for x in prior_rows[1 + k][1:]], comp[0], rtol=0, atol=0)
    summ = MO.summarize_node_ages(tr.ages()[:, 3, :], burn_in=0.25, names=[str(v) for v in range(topo.n_nodes)])
    assert np.all(summ.ci_lower <= summ.mean) and np.all(summ.mean <= summ.ci_upper) and np.all(summ.mean[topo.leaves] == 0)
    assert summ.render().count("\n") == topo.n_nodes + 1


def test_device_sampler_recovers_known_marginals(gpu):
    """Independent of the CPU twin: with a flat likelihood and the root-branch Jacobian lift switched off the chains
    sample the prior, whose marginals are known in closed form for the uncorrelated gamma clock: rVar ~ gamma(3/2, 1/6)
    (mean 1/4, variance 1/24) and rMu ~ exponential(1) (mean 1, variance 1); a soft calibration of the root keeps the
    prior of the time height proper.  A five-leaf tree with a brace lets all sixteen proposal kinds take part,
    so a wrong ratio or Jacobian in one of them shifts these moments.  The cycle is built with exact_jacobians=True: two
    Jacobians of the reference are not determinants (tests/test_mh_oracle.py::test_two_reference_jacobians_are_not_
    determinants) and, restated as they are, move E[rVar] to 0.24 and E[rate of a root child] to 1.03."""
    import dataclasses

    from mcmc_date_amd import monitor as MO

    # ((a,b),(c,(d,e))): both root children are inner nodes (pulley) and nodes 1 and 6 are braced
    parent = np.array([-1, 0, 1, 1, 0, 4, 4, 6, 6], np.int32)
    topo = M.Topology(parent)
    n = topo.n_nodes - 2
    lik = M.MvnLikelihood(M.Full(np.full(n, 0.5), np.eye(n) * 1e-12, 0.0)).bind_tree(topo)
    braces = [M.Brace("b", [1, 6], 0.05)]
    pf = M.PriorFunction(1.0, "UncorrelatedGamma", [M.Calibration("root", 0, 0.5, 0.025, 2.0, 0.025)], [], braces, topo)
    ps, missing = M.proposals(topo, braces, True, exact_jacobians=True)
    assert missing == []
    ps = [dataclasses.replace(p, jac_root=False) for p in ps]
    assert {p.kind for p in ps} == set(range(16))            # every proposal kind takes part
    B = 512
    smp = M.Sampler(lik, pf, ps, B, seed=314)
    smp.set_initial_state(M.State(1.0, 1.0, 1.0, np.array([1.0, 0.5, 0.0, 0.0, 0.8, 0.0, 0.45, 0.0, 0.0]), 1.0, 1.0,
                                  np.array([0.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0])))
    for period in (50, 50, 100, 100, 200, 200):
        smp.run(period)
        smp.autotune()
    tr = MO.collect(smp, 600, period=20)
    rvar, rmu = tr.rate_variance.ravel(), tr.rate_mean.ravel()
    n = rvar.size                                                # 30 samples x 512 chains, chains independent
    assert abs(rvar.mean() - 0.25) < 0.008 and abs(rvar.var() - 1.0 / 24.0) < 0.005, (rvar.mean(), rvar.var())
    assert abs(rmu.mean() - 1.0) < 0.04 and abs(rmu.var() - 1.0) < 0.15, (rmu.mean(), rmu.var())
    r_all = tr.rates[:, :, 1:]
    assert np.all(np.abs(r_all.mean(axis=(0, 1)) - 1.0) < 0.02), r_all.mean(axis=(0, 1))
    t, acc, tried = smp.tuning()
    rate = acc.sum(axis=0) / np.maximum(1, tried.sum(axis=0))
    assert np.all((rate > 0.05) & (rate < 0.95)), rate     # bounded slides on a three-leaf tree accept often


def test_heated_chains_and_mc3(gpu, golden):
    """`--mc3` of the reference (app/Main.hs:476-478): heated chains accept with (prior x likelihood)^beta -- step-level parity
    with the CPU twin under non-trivial temperatures -- and the swap phase on top (mcmc_date_amd.sampler.MC3): the cold
    chains of 16 groups of 4 reproduce the node ages of plain Metropolis-Hastings chains within 2 %."""
    fx = golden["12-leaves-variable-rate"]
    topo, ps, smp, twin = setup(fx, B=8, seed=31)
    beta = np.array([1.0, 0.9, 0.7, 0.5, 1.0, 0.97, 0.97 ** 2, 0.97 ** 3])
    smp.set_temperatures(beta)
    twin.beta[:] = beta
    sched = M.cycle_schedule(ps, 3, np.random.default_rng(1))
    tol = 1e-8 + 1e-12 * np.abs(smp.posterior()[:, :2]).max()
    ta, tk = smp.run_schedule(sched, trace=True)
    ra, rk = twin.run(sched, trace=True)
    fin = np.isfinite(ra)
    assert np.array_equal(np.isfinite(ta), fin) and alpha_close(ta[fin], ra[fin], tol) and np.array_equal(tk, rk)
    compare_states(smp, twin)
    with pytest.raises(M.McdError):
        smp.set_temperatures(np.array([1.0, 0.9, 0.7, 0.5, 1.0, 0.97, 0.0, 1.2]))
    # MC3 proper
    B = 64
    _, _, plain, _ = setup(fx, B=B, seed=41)
    _, _, heated, _ = setup(fx, B=B, seed=42)
    short = dict(fast=[10, 10, 20, 40, 80], slow=[100, 200, 300, 400])
    plain.burn_in(**short)
    plain.run(3000, accumulate=True)
    ages_plain = plain.node_age_summary()[0]
    heated.burn_in(**short)
    mc3 = M.MC3(heated, n_chains=4, swap_period=2, n_swaps=3, seed=5)
    assert len(mc3.cold()) == 16
    mc3.run(400)
    ages = mc3.run(3000, collect_ages=True)
    assert ages.shape == (1500, 16, topo.n_nodes) and len(mc3.cold()) == 16
    rate = mc3.swaps_accepted / np.maximum(1, mc3.swaps_tried)
    assert np.all((rate > 0.05) & (rate <= 1.0)), rate
    inner = ~topo.leaves
    rel = np.abs(ages.mean(axis=(0, 1))[inner] - ages_plain[inner]) / ages_plain[inner]
    assert rel.max() <= 0.02, (rel, rate)
    with pytest.raises(ValueError):
        M.MC3(heated, n_chains=5)
    with pytest.raises(ValueError):
        M.MC3(heated, n_chains=4, n_swaps=4)


def test_mc3_swap_phase_on_the_device(gpu, golden):
    """The swap phase behind the C ABI (mcd_mh_mc3_init / mcd_mh_mc3_swap / mcd_mh_mc3_get, csrc/k_mc3.hip) against its host
    restatement sampler.mc3_swap_host on the same counter-based draws: after every phase the temperature ranks of all chains, the
    temperatures the driver holds and the swap counters are the same, bit for bit (24 chains = 6 groups of 4, a steep ladder so
    that swaps are accepted and refused, 10 phases with two iterations of the cycle in between); a twin-backed MC3 (the CPU twin
    behind the same class) follows the device run phase by phase.  Structural faults return error codes."""
    import ctypes as C

    from twin_backend import TwinBackend

    fx = golden["12-leaves-variable-rate"]
    B = 24
    topo, ps, smp, twin = setup(fx, B=B, seed=51)
    ladder = np.array([1.0, 0.7, 0.45, 0.25])
    mc3 = M.MC3(smp, n_chains=4, swap_period=2, n_swaps=3, betas=ladder, seed=77)
    tb = TwinBackend(twin, ps, seed=51)
    ref = M.MC3(tb, n_chains=4, swap_period=2, n_swaps=3, betas=ladder, seed=77)
    lib = M._capi.lib()
    rank = np.arange(B, dtype=np.int32) % 4
    tried, acc = np.zeros(3, np.int64), np.zeros(3, np.int64)
    assert np.array_equal(mc3.rank, rank)
    for phase in range(10):
        smp.run(2)
        tb.run(2)
        post = smp.posterior()
        assert np.allclose(post, tb.posterior(), rtol=1e-12, atol=1e-7)
        M.sampler.mc3_swap_host(rank, post[:, 0] + post[:, 1], ladder, 3, mc3.seed, phase, tried, acc)
        mc3.swap()
        ref.swap()
        beta = np.empty(B)
        M._capi.check(lib.mcd_mh_mc3_get(smp._h, None, None, None, beta.ctypes.data_as(C.POINTER(C.c_double))))
        assert np.array_equal(mc3.rank, rank) and np.array_equal(beta, ladder[rank]), phase
        assert np.array_equal(mc3.swaps_tried, tried) and np.array_equal(mc3.swaps_accepted, acc)
        assert np.array_equal(ref.rank, rank) and np.array_equal(twin.beta, beta)          # the twin-backed class: same decisions
    assert acc.sum() > 0 and (tried - acc).sum() > 0 and not np.array_equal(rank, np.arange(B) % 4)
    assert sorted(rank[:4].tolist()) == [0, 1, 2, 3] and len(mc3.cold()) == B // 4
    compare_states(smp, twin)
    # faults
    assert lib.mcd_mh_mc3_init(smp._h, 5, ladder.ctypes.data_as(C.POINTER(C.c_double)), B, 1) == M._capi.MCD_ERR_INVALID_ARG     # 24 % 5
    assert lib.mcd_mh_mc3_init(smp._h, 4, ladder[::-1].copy().ctypes.data_as(C.POINTER(C.c_double)), B, 1) == M._capi.MCD_ERR_INVALID_ARG
    assert lib.mcd_mh_mc3_swap(smp._h, 4, None, 1, B) == M._capi.MCD_ERR_INVALID_ARG                                         # n_swaps > n - 1
    _, _, fresh, _ = setup(fx, B=8, seed=1)
    assert lib.mcd_mh_mc3_swap(fresh._h, 1, None, 1, 8) == M._capi.MCD_ERR_INVALID_ARG                                       # no init
    _, _, part, _ = setup(fx, B=8, seed=1, first_chain=8)
    assert lib.mcd_mh_mc3_init(part._h, 4, ladder.ctypes.data_as(C.POINTER(C.c_double)), 24, 1) == 0
    assert lib.mcd_mh_mc3_swap(part._h, 3, None, 1, 8) == M._capi.MCD_ERR_INVALID_ARG                                        # a shard needs the gathered values
    # a shard with the gathered posteriors of "three ranks": its own chains' temperatures follow the global table
    import torch
    g = torch.zeros((3, 3, 8), dtype=torch.float64, device=gpu)
    g[:, 1, :] = torch.as_tensor(np.random.default_rng(3).normal(size=(3, 8)) * 5.0, device=gpu)
    assert lib.mcd_mh_mc3_swap(part._h, 3, C.c_void_p(g.data_ptr()), 3, 8) == 0
    r_all = np.empty(24, np.int32)
    b_loc = np.empty(8)
    M._capi.check(lib.mcd_mh_mc3_get(part._h, r_all.ctypes.data_as(C.POINTER(C.c_int32)), None, None, b_loc.ctypes.data_as(C.POINTER(C.c_double))))
    expect = M.sampler.mc3_swap_host(np.arange(24, dtype=np.int32) % 4, g[:, 1, :].cpu().numpy().reshape(-1), ladder, 3, 1, 0)
    assert np.array_equal(r_all, expect) and np.array_equal(b_loc, ladder[expect[8:16]]) and not np.array_equal(expect, np.arange(24) % 4)


def test_cpp_sampler_mirror(gpu, golden, tmp_path):
    """The C++ host mirror (mcmc-date_amd/host/mcmcdate.hpp: priorFunction, initWith, proposals, Sampler) against the Python
    mirror: same proposal table, same initial prior / likelihood, and after the same schedule with the same seed the same
    posterior triples and states, bit for bit (both sit on the same C ABI)."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "test_host_mirror")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    fx = golden["24-leaves-braces"]
    B, seed = 4, 17
    topo, ps, smp, _ = setup(fx, B=B, seed=seed)
    sched = M.cycle_schedule(ps, 2, np.random.default_rng(3))
    r = lambda v: repr(float(v))
    p = tmp_path / "sampler.txt"
    with open(p, "w") as f:
        f.write(f"{len(fx['parent'])}\n" + " ".join(map(str, fx["parent"])) + "\n")
        f.write(" ".join(r(v) for v in fx["mu"]) + "\n" + " ".join(r(v) for v in fx["sigma_inv"].ravel()) + "\n" + r(fx["logdet"]) + "\n")
        f.write(f"{r(fx['prior_ht'])} 0 {len(fx['cal'])}\n")
        for c in fx["cal"]:
            f.write(f"{int(c[0])} {int(c[1])} {r(c[2])} {r(c[3])} {int(c[4])} {r(c[5])} {r(c[6])}\n")
        f.write(f"{len(fx['con'])}\n")
        for c in fx["con"]:
            f.write(f"{int(c[0])} {int(c[1])} {r(c[2])}\n")
        f.write(f"{len(fx['brace_sd'])}\n")
        for i, sd in enumerate(fx["brace_sd"]):
            nodes = fx["brace_nodes"][fx["brace_ptr"][i]:fx["brace_ptr"][i + 1]]
            f.write(f"{len(nodes)} {r(sd)} " + " ".join(str(int(v)) for v in nodes) + "\n")
        f.write(" ".join(r(v) for v in fx["mean_lengths"]) + "\n")
        f.write(f"{B} {seed} {sched.size}\n" + " ".join(str(int(v)) for v in sched.ravel()) + "\n")
    out = subprocess.run([exe, "--sampler", str(p)], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    tab = [int(v) for v in lines[0].split()[1:]]
    assert tab == [len(ps), sum(q.weight for q in ps), sum(q.kind for q in ps), sum(q.node for q in ps), sum(q.dim for q in ps),
                   M.weight_n_branches(topo.n_nodes)]
    post0 = smp.posterior()
    p0 = [float(v) for v in (lines[1].split()[1], lines[1].split()[3])]
    assert p0[0] == post0[0, 0] and p0[1] == post0[0, 1]
    smp.run_schedule(sched)
    post = smp.posterior()
    got = np.array([[float(v) for v in l.split()[2:]] for l in lines if l.startswith("post ")])
    assert np.array_equal(got, post)
    s = smp.state()
    st = [float(v) for v in [l for l in lines if l.startswith("state ")][0].split()[1:]]
    assert st == [s.time_height[B - 1], s.heights[B - 1, 1], s.rates[B - 1, topo.n_nodes - 1]]
    # the C++ mirror's Nuts (device NUTS through mcd_hmc_nuts_run) against the Python mirror on the same streams: bit for bit
    nl = [l for l in lines if l.startswith("nuts ")][0].split()[1:]
    lf = M.Leapfrog(smp._keep[0], smp._keep[1], len(fx["cal"]) > 0, B)
    lf.set_state(s)
    q0 = lf.position()[0]
    inv_mass = np.maximum((0.1 * np.abs(q0)).mean(axis=0) ** 2, 1e-12)
    eps, alpha, _, _ = lf.nuts_run(2, 0.02, inv_mass, adapt=False, max_depth=4, seed=seed)
    sn = lf.state()
    assert int(nl[0]) == lf.dim
    assert [float(v) for v in nl[1:]] == [alpha[0], alpha[B - 1], sn.time_height[B - 1], sn.heights[B - 1, 1]]
    # ... and its warm-up (mcd_hmc_nuts_warmup: step sizes and masses tuned in the library), continued from transition 2
    eps_w, im_w, _ = lf.nuts_warmup(eps, inv_mass, windows=1, window=6, delta=0.65, max_depth=4, seed=seed, first_transition=2)
    wl = [float(v) for v in [l for l in lines if l.startswith("warmup ")][0].split()[1:]]
    assert wl == [eps_w[0], eps_w[B - 1], im_w[0], im_w[lf.dim - 1]]
    # the swap phase of MC3 through the C++ mirror: two phases on the posteriors the sampler holds, groups of two chains
    ranks = np.arange(B, dtype=np.int32) % 2
    lnpi = post[:, 0] + post[:, 1]
    for phase in range(2):
        M.sampler.mc3_swap_host(ranks, lnpi, np.array([1.0, 0.3]), 1, 99, phase)
    assert [int(v) for v in [l for l in lines if l.startswith("mc3")][0].split()[1:]] == ranks.tolist()


def test_prior_only_node_ages_against_the_references_own_samples(gpu):
    """The one output of the reference itself that isolates rows f1 + f2 (prior x whole proposal cycle x Jacobians x root-branch
    lifts): the node ages of its six prior-only chains on the 7-taxon mtCDNApri data (`./run -s -f analysis.conf -c ul n r`,
    bench/comparison_with_mcmctree/README.md:615-632; summary statistics and the three input files in
    tests/golden/mtCDNApri_prior_samples.json, generator tests/golden/make_prior_sample_summary.py).  The device sampler
    runs the same analysis -- calibrations from the MCMCtree-style tree, uncorrelated log-normal clock, no likelihood, the
    reference's cycle, burn-in schedule and 8000 iterations -- with 128 chains.

    Round 2 reproduced the inner nodes and missed the root (22.4 against 19.0).  Round 3 found why
    (tests/test_reference_samples.py::test_the_samples_carry_a_root_bound_of_30, CPU): the committed samples carry a soft upper
    bound of the root at 30.0, the committed calibration tree says 100.  With the samples' bound ALL SIX node ages are within
    1 % of the reference's pooled means, their 2.5 / 97.5 % quantiles within 2 % (the lower quantile of the nodes 5 and 9, which
    reach down to 0.1, on the scale of the mean).  With the committed bound the root's mean is 22.4 as before, and the same
    samples cut at 30.4 have the reference's means.  The reference's own Jacobians fit its samples better than the determinants
    (`exact_jacobians=True`)."""
    import mtcdnapri as A
    from mcmc_date_amd import monitor as MO

    fx = A.golden("prior")
    ref = {k: np.array(v) for k, v in fx["pooled"].items()}
    dev = {}
    for upper, exact in ((A.ROOT_UPPER_OF_THE_SAMPLES, False), (A.ROOT_UPPER_OF_THE_SAMPLES, True), (None, False)):
        an = A.analysis("NoLikelihood", root_upper=upper, exact_jacobians=exact)
        lik = M.MvnLikelihood(M.Full(an.mu, an.sigma_inv, an.logdet)).bind_tree(an.topo)          # NoData: likelihood 1
        pf = M.PriorFunction(an.ht, "UncorrelatedLogNormal", an.cal, [], [], an.topo)
        smp = M.Sampler(lik, pf, an.table, 128, seed=11 + int(exact))
        smp.set_initial_state(M.init_with(an.topo, an.prep.mean_lengths))   # initWith: time height 1.0, as the reference starts
        smp.burn_in()
        tr = MO.collect(smp, 8000, period=20)
        ages = tr.ages()[:, :, fx["nodes"]].reshape(-1, len(fx["nodes"]))
        mean = ages.mean(axis=0)
        q = np.quantile(ages, [0.025, 0.975], axis=0)
        dev[(upper, exact)] = np.abs(mean - ref["mean"]) / ref["mean"]
        if upper is not None and not exact:
            assert np.all(dev[(upper, exact)] <= 0.01), (mean, ref["mean"])
            assert np.all(np.abs(q[0] - ref["q025"]) <= 0.02 * ref["mean"]), (q[0], ref["q025"])
            assert np.all(np.abs(q[1] - ref["q975"]) <= 0.02 * ref["q975"]), (q[1], ref["q975"])
            assert np.all(np.abs(ages.std(axis=0) - ref["sd"]) <= 0.03 * ref["sd"])
            assert np.max(np.abs(np.corrcoef(ages.T) - np.array(fx["correlation"]))) <= 0.03
        if upper is None:                                                   # the calibration tree as committed
            assert 1.12 * ref["mean"][0] < mean[0] < 1.25 * ref["mean"][0], mean
            cut = ages[ages[:, 0] < 30.4]
            assert np.all(np.abs(cut.mean(axis=0) - ref["mean"]) <= 0.01 * ref["mean"]), cut.mean(axis=0)
    b = A.ROOT_UPPER_OF_THE_SAMPLES
    assert np.all(dev[(b, False)] <= dev[(b, True)] + 0.003), dev                   # the reference's samples side with its own Jacobians
    assert dev[(b, True)][0] > 0.01, dev                                            # the determinants move the root by more than 1 %


@pytest.mark.parametrize("native_sparse", [False, True])
def test_posterior_node_ages_against_the_references_own_samples(gpu, tmp_path, native_sparse):
    """The reference's own POSTERIOR output pins the whole path -- prepare (with the graphical lasso), likelihood, prior, proposal
    cycle, Jacobians: the node ages of its six chains WITH data on the 7-taxon mtCDNApri analysis
    (`./run -s -f analysis.conf -c ul s r`: SparseMultivariateNormal 0.1, bench/comparison_with_mcmctree/README.md:615-632;
    summary statistics in tests/golden/mtCDNApri_post_samples.json, generator make_post_sample_summary.py; the inputs are in the
    prior-only fixture).  The device sampler runs the same analysis with 128 chains: every node age -- the root included --
    within 1 % of the reference's pooled mean (north_star's bar; measured 0.1 .. 0.5 %), the 2.5 % / 97.5 % quantiles within 3 %
    (measured up to 1.7 %).  With `exact_jacobians=True` the root's upper quantile drifts by 2.9 %: the reference's samples side
    with its own Jacobians here as in the prior-only runs (tools/post_samples_check.py, profiles/r02_post_samples_variants.jsonl)."""
    import json
    import os

    from mcmc_date_amd import monitor as MO
    from mcmc_date_amd.prepare import prepare

    here = os.path.join(os.path.dirname(__file__), "golden")
    fx = json.load(open(os.path.join(here, "mtCDNApri_prior_samples.json")))
    post = json.load(open(os.path.join(here, "mtCDNApri_post_samples.json")))
    paths = {}
    for k in ("rooted_tree", "calibration_tree", "tree_list"):
        paths[k] = str(tmp_path / k)
        open(paths[k], "w").write(fx["inputs"][k])
    prep = prepare(paths["tree_list"], paths["rooted_tree"], "SparseMultivariateNormal 0.1")
    assert isinstance(prep.lhd, M.Sparse) and len(prep.lhd.sigma_inv_assoc) < len(prep.mu) ** 2      # really sparse
    topo = prep.topology
    cal = M.load_calibrations_from_tree(topo, paths["calibration_tree"])
    ht = M.get_mean_root_height(cal)
    # native_sparse: the Sparse record as it is -- the precision matrix in CSR on the device, mcd_mh_create_sparse, every proposal inside a
    # segment launch (k_mh_segment_sparse.hip: 13 nodes, all 11 distances fit the list); else the densified record through the dense handle
    lik = (M.SparseLikelihood(prep.lhd) if native_sparse else M.MvnLikelihood(prep.lhd)).bind_tree(topo)
    pf = M.PriorFunction(ht, "UncorrelatedLogNormal", cal, [], [], topo)
    ps, missing = M.proposals(topo, [], calibrations_available=True)
    assert missing == []
    smp = M.Sampler(lik, pf, ps, 128, seed=21)
    x0 = M.init_with(topo, prep.mean_lengths)
    x0.time_height = ht
    smp.set_initial_state(x0)
    smp.burn_in()
    tr = MO.collect(smp, 8000, period=20)
    assert ("segments over a sparse precision matrix" in smp.last_path()) == native_sparse
    ages = tr.ages()[:, :, post["nodes"]].reshape(-1, len(post["nodes"]))
    ref = {k: np.array(v) for k, v in post["pooled"].items()}
    dev = np.abs(ages.mean(axis=0) - ref["mean"]) / ref["mean"]
    assert np.all(dev <= 0.01), dev
    q = np.quantile(ages, [0.025, 0.975], axis=0)
    assert np.all(np.abs(q[0] - ref["q025"]) <= 0.03 * ref["q025"]), (q[0], ref["q025"])
    assert np.all(np.abs(q[1] - ref["q975"]) <= 0.03 * ref["q975"]), (q[1], ref["q975"])


def test_shard_allgather_through_rccl(gpu, golden):
    """The path's one exchange step on hardware: the C ABI's all-gather (mcd_shard_*: RCCL loaded at run time,
    communicator made from a unique id, ncclAllGather on the sampler's stream) of the device-resident per-chain ln posterior of
    a sampler -- with the one rank a one-GPU box allows, so that the RCCL code path has run once: the gathered values are
    the sampler's own.  The same through torch.distributed's nccl backend (what shards.py and bench.py --gpus N use).  The
    N > 1 logic (ragged shards, global chain order, the same chains whatever the number of ranks) is covered by the
    world-size-2 gloo tests (tests/test_shards_gloo.py)."""
    import ctypes as C
    import os

    import torch
    import torch.distributed as dist

    from mcmc_date_amd import shards

    topo, ps, smp, twin = setup(golden["12-leaves-variable-rate"], B=64, seed=5)
    smp.run(2)
    post = smp.posterior()                                              # host copy [B, 3]
    lib = M._capi.lib()
    dptr, st = C.c_void_p(), C.c_void_p()
    M._capi.check(lib.mcd_mh_posterior_device(smp._h, C.byref(dptr), C.byref(st)))
    comm = shards.ShardComm(shards.ChainShard(0, 1, 64))
    send = torch.empty(3 * 64, dtype=torch.float64, device=gpu)
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(send.data_ptr()), dptr, 3 * 64 * 8, 3)   # device to device
    out = comm.allgather(send)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(3, 64).T, post)
    # straight from the sampler's device array, on the sampler's stream
    out2 = torch.zeros(3 * 64, dtype=torch.float64, device=gpu)
    M._capi.check(lib.mcd_shard_allgather(comm._comm, dptr, C.c_void_p(out2.data_ptr()), 3 * 64, st))
    torch.cuda.synchronize()
    assert torch.equal(out2, out)
    comm.close()
    assert lib.mcd_shard_allgather(None, dptr, C.c_void_p(out2.data_ptr()), 1, None) == M._capi.MCD_ERR_INVALID_ARG
    # torch.distributed over RCCL, one rank
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
    try:
        g = torch.empty(64, dtype=torch.float64, device=gpu)
        dist.all_gather_into_tensor(g, send[64:128].contiguous())
        torch.cuda.synchronize()
        assert np.array_equal(g.cpu().numpy(), post[:, 1])
        assert torch.equal(shards.gather_loglik(send[64:128], shards.ChainShard(0, 1, 64)), send[64:128])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("knob", ["MCD_MH_PRIOR_WAVES", "MCD_MH_SEG_TAIL", "MCD_MH_PRIOR_DRAWS"])
@pytest.mark.parametrize("n_leaves,B,sparse", [(200, 33, False), (513, 16, False), (7, 8, True), (300, 17, True), (1007, 5, True)])
def test_prior_waves_of_the_segment_kernels_give_the_same_chains(gpu, n_leaves, B, sparse, knob, knobs):
    """The segment kernels (k_mh_segment.hip, k_mh_segment_sparse.hip) give every chain two PRIOR waves beside its chain wave and its
    likelihood wave: the birth-death and the clock block of a proposal's ln prior are evaluated by them (mh_segment_device.hpp:
    seg_prior_wave) while the chain wave evaluates the node priors.  The same functions on the same numbers in the same order: with the
    knob MCD_MH_PRIOR_WAVES = 0 (the chain wave evaluates all three blocks, round 3's arrangement) every ln acceptance ratio, decision,
    state and posterior term is the same bits.  Calibrations and a constraint, so that all three blocks are live; an odd batch.

    The same for MCD_MH_SEG_TAIL: a dense proposal that follows a segment is proposed by that segment's launch from the state it holds in
    LDS (mh_segment_device.hpp: MhSegPending::p_tail, seg_tail_distances) -- k_mh_step_wg's proposal half without the launch and without
    reading the state back; 0 = by the step kernel."""
    from mcmc_date_amd import synthetic as S

    topo = S.random_topology(n_leaves, seed=31)
    n = topo.n_nodes - 2
    if sparse:
        _, assoc = S.banded_precision(n, seed=n)
        lik = M.SparseLikelihood(M.Sparse(np.random.default_rng(1).uniform(0.01, 0.2, n), assoc, 0.0)).bind_tree(topo)
    else:
        mu, sigma = S.random_spd_problem(n, seed=n)
        lik = M.MvnLikelihood.from_covariance(mu, sigma).bind_tree(topo)
    inner = [v for v in range(1, topo.n_nodes) if (np.asarray(topo.parent) == v).any()]
    cal = [M.Calibration("root", 0, 0.9, 0.025, 1.3, 0.025), M.Calibration("c", int(inner[len(inner) // 2]), 1e-3, 0.025, 5.0, 0.025)]
    con = [M.Constraint("k", int(inner[-1]), int(topo.parent[inner[-1]]), 0.025)] if topo.parent[inner[-1]] > 0 else []
    pf = M.PriorFunction(1.0, "UncorrelatedLogNormal", cal, con, [], topo)
    ps, _ = M.proposals(topo, [], calibrations_available=True)
    s0 = S.random_states(topo, B, seed=5)
    s0.time_birth_rate = np.full(B, 1.0); s0.time_death_rate = np.full(B, 0.8); s0.rate_variance = np.full(B, 0.3)
    cyc = M.cycle_schedule(ps, 1, np.random.default_rng(2))
    sched = np.tile(cyc, (1, 700 // cyc.shape[1] + 1))[:, :700]
    out = {}
    for waves in ("1", "0"):
        knobs.setenv(knob, waves)
        smp = M.Sampler(lik, pf, ps, B, seed=77)
        smp.set_state(s0)
        ta, tk = smp.run_schedule(sched, trace=True)
        assert "segments" in smp.last_path()
        out[waves] = (ta, tk, smp.state(), smp.posterior())
    a, b = out["1"], out["0"]
    assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1]) and 0.02 < a[1].mean() < 0.98
    for f in ("time_birth_rate", "time_death_rate", "time_height", "heights", "rate_mean", "rate_variance", "rates"):
        assert np.array_equal(getattr(a[2], f), getattr(b[2], f)), f
    assert np.array_equal(a[3], b[3])


@pytest.mark.parametrize("sparse", [False, True])
def test_dense_proposals_at_every_position_of_a_segmented_run(gpu, sparse, knobs):
    """A schedule made by hand around the places where the segment path changes hands (mh_capi.cpp): a dense proposal (one that moves every
    distance: the scalings of the time height and of the rate mean) as the run's first step, directly after another dense one, as the step a
    recomputation of z / q falls on (every 256th), as the step after it, and as the run's last step -- proposed by the preceding segment's
    launch where there is one (MhSegPending::p_tail), by the step kernel otherwise, decided by the following segment's launch or by the step
    kernel.  Every ln acceptance ratio, decision, state and posterior term equals the run with MCD_MH_SEG_TAIL = 0 bit for bit, and the run
    with two launches per step (MCD_MH_SEGMENTS = 0) in its decisions and states."""
    from mcmc_date_amd import synthetic as S
    from mcmc_date_amd import sampler as SM

    topo = S.random_topology(180, seed=9)
    n = topo.n_nodes - 2
    B = 7
    if sparse:
        _, assoc = S.banded_precision(n, seed=n)
        lik = M.SparseLikelihood(M.Sparse(np.random.default_rng(1).uniform(0.01, 0.2, n), assoc, 0.0)).bind_tree(topo)
    else:
        mu, sigma = S.random_spd_problem(n, seed=n)
        lik = M.MvnLikelihood.from_covariance(mu, sigma).bind_tree(topo)
    pf = M.PriorFunction(1.0, "UncorrelatedGamma", [M.Calibration("root", 0, 0.9, 0.025, 1.3, 0.025)], [], [], topo)
    ps, _ = M.proposals(topo, [], calibrations_available=True)
    tab = M.table_arrays(ps)
    dense = [i for i in range(len(ps)) if tab["kind"][i] == SM.SCALE_SCALAR and tab["node"][i] in (SM.TIME_HEIGHT, SM.RATE_MEAN)]
    few = [i for i in range(len(ps)) if tab["kind"][i] in (SM.SLIDE_NODE, SM.SCALE_BRANCH_RATE)]
    assert len(dense) == 2 and len(few) > 100
    rng = np.random.default_rng(4)
    sched = rng.choice(few, size=600).astype(np.int32)
    for pos, d in ((0, 0), (40, 0), (41, 1), (42, 0), (255, 1), (256, 0), (257, 1), (300, 0), (511, 0), (512, 1), (599, 1)):
        sched[pos] = dense[d]
    sched = sched[None, :]
    s0 = S.random_states(topo, B, seed=5)
    s0.time_birth_rate = np.full(B, 1.0); s0.time_death_rate = np.full(B, 0.8); s0.rate_variance = np.full(B, 0.3)
    out = {}
    for name, knob, val in (("tail", None, None), ("step kernel", "MCD_MH_SEG_TAIL", "0"), ("two launches", "MCD_MH_SEGMENTS", "0")):
        if knob:
            knobs.setenv(knob, val)
        smp = M.Sampler(lik, pf, ps, B, seed=21)
        smp.set_state(s0)
        ta, tk = smp.run_schedule(sched, trace=True)
        ta2, tk2 = smp.run_schedule(sched[:, :301], trace=True)      # (a second run: starts on a dense proposal, ends on one)
        assert ("segments" in smp.last_path()) == (name != "two launches"), smp.last_path()
        out[name] = (np.concatenate([ta, ta2]), np.concatenate([tk, tk2]), smp.state(), smp.posterior())
        if knob:
            knobs.delenv(knob)
    a, b, c = out["tail"], out["step kernel"], out["two launches"]
    assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
    assert 0.02 < a[1].mean() < 0.98 and a[1][[0, 40, 41, 256, 599]].any()
    assert np.array_equal(a[1], c[1])
    np.testing.assert_allclose(a[0], c[0], rtol=0, atol=1e-6)
    for f in ("time_birth_rate", "time_death_rate", "time_height", "heights", "rate_mean", "rate_variance", "rates"):
        assert np.array_equal(getattr(a[2], f), getattr(b[2], f)), f
        assert np.array_equal(getattr(a[2], f), getattr(c[2], f)), f
