#!/usr/bin/env python3
"""bench.py -- MVN log-likelihood evals/sec at N = 256 (BASELINE.json metric), one process per GPU.

A "step" is one pass of the hot path over one batch of chains: one launch of the batched
log-likelihood kernel over `--chains` chains (default 512, BASELINE.json configs[2]: synthetic
256-dimensional problem, dense random Sigma, 512 chains on one MI355X).  Inputs are resident in
HBM before the timed region.  With --gpus N > 1 (torchrun, RCCL) every rank evaluates its own 512
chains (weak scaling; the path has no exchange step -- chains are independent).  `--swap-period P`
adds the sampler-level exchange of config 5 (an all-gather of the per-chain log-likelihoods every
P steps, mirroring MC3's SwapPeriod, app/Main.hs:477); it is off by default because it is not part
of the likelihood path.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel against HBM (north_star);
`cpu_baseline` times the CPU oracle (a restatement of the reference's algebra -- the Haskell
toolchain is absent) on a bounded sample on rank 0's host core.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # vector fp64 FMA peak (256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz)


def algorithmic_bytes_per_eval(n: int, batch: int) -> float:
    """x (8n) + ll (8) + (mu (8n) + packed L (4n(n+1))) amortised over the launch."""
    return 8.0 * n + 8.0 + (8.0 * n + 4.0 * n * (n + 1)) / batch


def algorithmic_flops_per_eval(n: int) -> float:
    return float(n) * n + 5.0 * n


def measured_traffic(key=None):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
    passes, gfx950 correction applied): profiles/*_pmc_traffic.json of the latest round, entry `key` (the workload the passes
    were run on: "n256", "n1024", "tree255", "tree1023" -- tools/collect_profiles_r02.sh), or None."""
    import glob

    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not fs:
        return None, None
    try:
        d = json.load(open(fs[-1]))
        src = "profiles/" + os.path.basename(fs[-1]) + (f"[{key}]" if key is not None else "") + " (committed rocprofv3 --pmc passes, not collected by this run)"
        if key is not None:
            return float(d[key]["per_launch_bytes_corrected"]), src
        return float(d["per_launch_bytes_corrected"]), src
    except Exception:
        return None, None


def cpu_baseline(n, mu, sigma, X, budget_s=12.0):
    """The oracle in the reference's own algebra (Sigma^-1 dgemv + ddot, one evaluation per call,
    fresh dx buffer per call -- app/Probability.hs:167-173) on one host core."""
    import oracle as O

    P = np.linalg.inv(sigma)
    logdet = float(np.linalg.slogdet(sigma)[1])
    O.build(native=True, force=True)   # -march=native: always rebuild on the host that runs it
    sample = X[: min(len(X), 512)]
    O.logpdf_full_batch(mu, P, logdet, sample[:32], native=True)  # warm
    t0 = time.perf_counter()
    evals = 0
    while True:
        O.logpdf_full_batch(mu, P, logdet, sample, native=True)
        evals += len(sample)
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    out = {"value": evals / dt, "unit": "evals/s", "cores": 1, "kind": "port",
           "sample": f"{evals} evaluations (sweeps over {len(sample)} of the bench's chains, n={n}) in {dt:.1f} s; "
                     "C restatement of app/Probability.hs:167-173 (Sigma^-1 form), gcc -O3 -march=native"}
    # the same port with the chains split statically over the host cores this process may use
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    O.logpdf_full_batch(mu, P, logdet, sample, native=True, all_cores=True)
    t0 = time.perf_counter()
    evals_mt = 0
    while time.perf_counter() - t0 < 4.0:
        O.logpdf_full_batch(mu, P, logdet, sample, native=True, all_cores=True)
        evals_mt += len(sample)
    out["all_cores"] = {"value": evals_mt / (time.perf_counter() - t0), "unit": "evals/s", "cores": threads}
    out["haskell_toolchain"] = haskell_toolchain()          # `kind` could only be "haskell" with one (and the reference's sources)
    return out


def mh_measure(dev_index, n, B, steps, warm, seed=3, rank=0, world=1, swap_period=0, swap_steps=0, rehearsal=False, sparse=False, repeats=1,
               tune_periods=0):
    """Lock-step Metropolis-Hastings on the device (the metric's "= MCMC steps/sec x chains" reading):
    a synthetic tree of dimension n (255 for --n 256: 2L - 3 is odd), the reference's whole proposal cycle
    (app/Definitions.hs:127-278) in its shuffled order, B chains per GPU stepping together; one step = one proposal of the cycle,
    evaluated (prior + likelihood), accepted or rejected in every chain.  With world > 1 the chains are a global set of
    world x B, this rank holding [rank B, (rank + 1) B) (shards.shard_sampler: global chain index = random-stream id).
    swap_period = P > 0 (BASELINE.json config 5; `mc3 (MC3Settings (NChains 4) (SwapPeriod P) (NSwaps 3))`, app/Main.hs:476-478):
    every P iterations of the cycle (or every `swap_steps` lock steps, for rehearsals shorter than an iteration) the ranks
    all-gather their [3][B] ln posteriors on the sampler's stream (mcd_shard_allgather: RCCL over xGMI) and every rank runs the
    swap phase of all groups on the gathered values (mcd_mh_mc3_swap) -- inside the timed region.  tune_periods = T > 0: before the
    warm-up, T periods of one iteration of the cycle each followed by `mcmc`'s auto-tuning (mcd_mh_tune) -- the reference always samples
    with tuned proposals (burnIn = BurnInWithCustomAutoTuning, app/Definitions.hs:420-424); the untuned default accepts most proposals,
    which is the expensive case for the persistent kernels (a proposal drawn ahead is thrown away after an acceptance).  The line says
    which acceptance rate the timed steps had.  Returns a dict for the JSON line."""
    import torch

    import mcmc_date_amd as M
    from mcmc_date_amd import shards as SH
    from mcmc_date_amd import synthetic as S

    topo = S.random_topology((n + 3) // 2, seed=seed)
    nd = topo.n_nodes - 2
    if sparse:
        # the reference's production configuration: the precision matrix kept sparse (band 3 + 4 random entries per row: the density of a
        # graphical-lasso estimate), mcd_mh_create_sparse
        _, assoc = S.banded_precision(nd, seed, 3, 4)
        tl = M.SparseLikelihood(M.Sparse(np.random.default_rng(seed).uniform(0.01, 0.2, nd), assoc, 0.0), device=dev_index).bind_tree(topo)
    else:
        mu, sigma = S.random_spd_problem(nd, seed=seed)
        tl = M.MvnLikelihood.from_covariance(mu, sigma, device=dev_index).bind_tree(topo)
    pf = M.PriorFunction(1.0, "UncorrelatedGamma", [], [], [], topo, device=dev_index)
    ps, _ = M.proposals(topo, [], calibrations_available=True)
    shard = SH.ChainShard(rank, world, world * B)
    s0 = S.random_states(topo, B, seed=seed + 1 + 1000 * rank)
    s0.time_birth_rate = np.full(B, 1.0)
    s0.time_death_rate = np.full(B, 0.8)
    s0.rate_variance = np.full(B, 0.3)
    smp = SH.shard_sampler(tl, pf, ps, world * B, 13, shard)
    smp.set_state(s0)
    S_iter = int(sum(p.weight for p in ps))
    cyc = M.cycle_schedule(ps, 1, np.random.default_rng(0))
    reps = max(1, (repeats * steps + warm) // cyc.shape[1] + 1)
    sched = np.tile(cyc, (1, reps))
    mc3, comm, swap_info = None, None, None
    period = 0
    if swap_period > 0 or swap_steps > 0:
        period = int(swap_steps) if swap_steps > 0 else int(swap_period) * S_iter
        if world > 1 and not rehearsal:
            comm = SH.ShardComm(shard, device=dev_index)
            mc3 = SH.mc3_for_shard(smp, shard, comm, n_chains=4, swap_period=max(1, swap_period), n_swaps=3, seed=7)
        elif world > 1:
            # one-GPU rehearsal (every rank on cuda:0, RCCL refuses that): the gather goes through the host and gloo
            def gather(sampler):
                local = np.ascontiguousarray(sampler.posterior().T)
                return torch.as_tensor(SH.gather_posterior_host(local, shard), device=torch.device("cuda", dev_index))

            mc3 = M.MC3(smp, n_chains=4, swap_period=max(1, swap_period), n_swaps=3, seed=7, shard=shard, gather=gather)
        else:
            mc3 = M.MC3(smp, n_chains=4, swap_period=max(1, swap_period), n_swaps=3, seed=7)

    phases = [0]

    def advance(lo, hi):
        """lock steps [lo, hi) of the schedule; a swap phase after every full period"""
        pos = lo
        while pos < hi:
            nxt = hi if period == 0 else min(hi, (pos // period + 1) * period)
            smp.run_schedule(sched[:, pos:nxt])
            if period and nxt % period == 0:
                mc3.swap()
                phases[0] += 1
            pos = nxt

    for _ in range(int(tune_periods)):
        smp.run_schedule(M.cycle_schedule(ps, 1, np.random.default_rng(100 + _)))
        smp.autotune()
    advance(0, warm)
    smp.reset_counters()
    if mc3 is not None and warm < period:                      # the first phase of a short run: exercised once outside the clock
        mc3.swap()
    torch.cuda.synchronize()
    phases[0] = 0
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    t0 = time.perf_counter()
    advance(warm, warm + steps)
    if mc3 is not None and phases[0] == 0:                     # fewer steps than a period: the exchange still runs once under the clock
        mc3.swap()
        phases[0] += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # (secondary fields of the default line only: a second pass over the same steps, the faster of the two -- on this pool about one run
    # in twenty sees the host learn of a completion some 50 ms late, which a 35 ms measurement cannot absorb; the headline kinds time
    # exactly the K steps asked for)
    for _ in range(max(0, repeats - 1)):
        t0 = time.perf_counter()
        advance(warm + steps, warm + 2 * steps)
        torch.cuda.synchronize()
        dt = min(dt, time.perf_counter() - t0)
    _, n_acc, n_try = smp.tuning()
    post = smp.posterior()
    assert np.all(np.isfinite(post)), "non-finite ln posterior after the Metropolis-Hastings run"
    if mc3 is not None:
        rk = mc3.rank
        assert all(sorted(rk[g * 4:(g + 1) * 4].tolist()) == [0, 1, 2, 3] for g in range(len(rk) // 4)), "temperature ranks are not a permutation"
        swap_info = {"n_chains": 4, "n_swaps": 3, "period_iterations": int(swap_period), "period_lock_steps": int(period), "phases_timed": int(phases[0]),
                     "ranks": int(world), "bytes_gathered_per_phase": int(world * 3 * B * 8),
                     "allgather": ("mcd_shard_allgather (RCCL ncclAllGather) on the sampler's stream" if comm is not None else
                                   "gloo through the host (one-GPU rehearsal)" if world > 1 else "none (one rank: the sampler's own array)"),
                     "swaps_tried": mc3.swaps_tried.tolist(), "swaps_accepted": mc3.swaps_accepted.tolist()}
    out = {"value": B * steps / dt, "unit": "proposal steps/s (lock steps x chains)", "us_per_lockstep": 1e6 * dt / steps,
           "n_nodes": int(topo.n_nodes), "dimension": int(nd), "chains": int(B), "lock_steps": int(steps),
           "proposals_per_iteration": S_iter, "likelihood": "sparse precision matrix (CSR on the device)" if sparse else "dense factor",
           "lds_bytes_per_workgroup": smp.last_dynamic_lds(),
           "acceptance_rate": float(n_acc.sum()) / max(1.0, float(n_try.sum())), "tune_periods": int(tune_periods),
           "what": "reference proposal cycle (16 kinds), prior + likelihood + accept/reject on the device; " + smp.last_path()}
    if swap_info:
        out["mc3"] = swap_info
    if comm is not None:
        out["rccl_comm_ranks"] = comm.count()
        comm.close()
    return out


def full_gpu_measure(dev_index, n, chains=8192, launches=200):
    """What the GPU does when it is FULL: the same log-density at `chains` chains, where every SIMD has work and the launch takes the
    multiply form on the fp64 matrix cores (k_wide.hip) -- priced against the dense fp64 MFMA peak.  A secondary
    field of the default line (the headline stays BASELINE.json's 512 chains, a latency-bound launch)."""
    import torch

    import mcmc_date_amd as M
    from mcmc_date_amd import synthetic as S

    dev = torch.device("cuda", dev_index)
    mu, sigma = S.random_spd_problem(n, seed=n)
    lik = M.MvnLikelihood.from_covariance(mu, sigma, device=dev_index)
    X = torch.as_tensor(S.sample_chains(mu, sigma, chains, seed=n + 77), device=dev)
    ll = torch.empty(chains, dtype=torch.float64, device=dev)
    for _ in range(20):
        lik.logpdf_into(X, ll)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        lik.logpdf_into(X, ll)
    e1.record()
    torch.cuda.synchronize()
    per = e0.elapsed_time(e1) * 1e-3 / launches
    chk = M.MvnLikelihood.from_covariance(mu, sigma, device=dev_index)
    chk.set_form("sweep")
    ref = chk.logpdf(X[:256])
    assert float(((ll[:256] - ref).abs() / ref.abs()).max()) <= 1e-11, "full-GPU batch differs from the sweep form"
    flops = algorithmic_flops_per_eval(n) * chains / per / 1e12
    hbm = algorithmic_bytes_per_eval(n, chains) * chains / per / 1e9
    return {"value": chains / per, "unit": "evals/s", "chains": int(chains), "n": int(n), "kernel_us_per_launch": per * 1e6, "launches": int(launches),
            "launch": "eager", "form": "multiply (v_mfma_f64_16x16x4_f64)",
            "roofline": {"bound": "mfma", "achieved": flops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / FP64_PEAK_TFLOPS,
                         "hbm_gbs": hbm, "hbm_frac": hbm / HBM_PEAK_GBS}}


def sparse_measure(dev_index, n, B, steps, warm, band=3, extra=4):
    """The sparse form (csrc/k_sparse.hip; logDensitySparseMultivariateNormal, app/Probability.hs:178-184): a synthetic symmetric,
    diagonally dominant precision matrix with a band and `extra` random entries per row -- the density a graphical-lasso estimate of a
    large tree has -- B chains, device resident.  HBM-priced: the CSR stream (12 bytes per nonzero) once per launch plus the chain vectors."""
    import torch

    import mcmc_date_amd as M
    from mcmc_date_amd import synthetic as S

    rng = np.random.default_rng(n)
    P, assoc = S.banded_precision(n, n, band, extra)
    mu = rng.uniform(0.01, 0.2, n)
    sp = M.SparseLikelihood(M.Sparse(mu, assoc, 0.0), device=dev_index)
    dev = torch.device("cuda", dev_index)
    X = torch.as_tensor(mu + 0.01 * rng.standard_normal((B, n)), device=dev)
    ll = torch.empty(B, dtype=torch.float64, device=dev)
    lib = M._capi.lib()

    def step():
        M._capi.check(lib.mcd_sparse_logpdf_batch(sp._h, X.data_ptr(), X.stride(0), B, 1, torch.cuda.current_stream().cuda_stream, ll.data_ptr()))

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = e0.elapsed_time(e1) * 1e-3 / steps
    dx = X.cpu().numpy() - mu
    Pc = P.tocsr()
    ref = -n * 0.9189385332046727 - 0.5 * np.einsum("bi,bi->b", dx, (Pc @ dx.T).T)
    err = float(np.max(np.abs(ll.cpu().numpy() - ref) / np.abs(ref)))
    assert err <= 1e-12, f"sparse form differs from scipy.sparse: {err}"
    alg = 12.0 * sp.nnz + 4.0 * (n + 1) + 8.0 * n + B * (8.0 * n + 8.0)
    flops = (2.0 * sp.nnz + 3.0 * n) * B / per / 1e12
    return dt, per, {"n": int(n), "nnz": int(sp.nnz), "chains": int(B), "chains_per_tile": int({True: 16}.get(False, 0)) or None,
                     "kernel_us_per_launch": per * 1e6, "alg_bytes_per_launch": alg, "fp64_tflops": flops,
                     "roofline": {"bound": "hbm", "achieved": alg / per / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / per / 1e9 / HBM_PEAK_GBS,
                                  "traffic": measured_traffic(f"sparse_{n}x{B}")[0], "traffic_source": measured_traffic(f"sparse_{n}x{B}")[1]}}


def haskell_toolchain():
    """is a Haskell toolchain on this box?  (`cpu_baseline.kind` says "haskell" only if the reference binary
    could be built and timed; the reference's sources are not on the GPU box either, so a toolchain alone changes nothing -- reported.)"""
    import shutil

    return {t: shutil.which(t) for t in ("ghc", "cabal", "stack")}


def e2e_measure(dev_index, chains=128, iterations=8000, period=2, seed=21, cpu_budget_s=15.0):
    """The reference's ONLY published timing, end to end: the posterior analysis of the 7-taxon mtCDNApri data --
    `./run -s -f analysis.conf -c ul s r`: sparse multivariate normal likelihood (graphical lasso 0.1), uncorrelated log-normal clock,
    calibrations from the MCMCtree-style tree, the burn-in schedule with auto tuning (app/Definitions.hs:420-424) + 8000 iterations
    (:440-441), monitors with period 2 (bench/comparison_with_mcmctree/README.md:615-632) -- 154 s per chain on an i7-1165G7 (README.md:720;
    other hardware: a stated baseline, not a ratio to claim).  Here: `prepare` on the host (timed apart), then `chains` chains in lock
    step on the device through the NATIVE sparse handle (mcd_mh_create_sparse: every proposal inside a segment launch), the chains'
    states fetched every 2 iterations as the monitors would; the node ages against the reference's committed samples
    (tests/golden/mtCDNApri_post_samples.json; must stay within 1 %).  Beside it the CPU twin (oracle/mh_oracle.c, a restatement) timed on
    a bounded sample on this box's host cores."""
    import tempfile

    import mcmc_date_amd as M
    from mcmc_date_amd import monitor as MO
    from mcmc_date_amd.prepare import prepare

    gold = os.path.join(ROOT, "tests", "golden")
    fx = json.load(open(os.path.join(gold, "mtCDNApri_prior_samples.json")))
    post = json.load(open(os.path.join(gold, "mtCDNApri_post_samples.json")))
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory() as d:
        paths = {}
        for k in ("rooted_tree", "calibration_tree", "tree_list"):
            paths[k] = os.path.join(d, k)
            open(paths[k], "w").write(fx["inputs"][k])
        prep = prepare(paths["tree_list"], paths["rooted_tree"], "SparseMultivariateNormal 0.1")
        topo = prep.topology
        cal = M.load_calibrations_from_tree(topo, paths["calibration_tree"])
    prepare_s = time.perf_counter() - t0
    ht = M.get_mean_root_height(cal)
    ps, _ = M.proposals(topo, [], calibrations_available=True)
    S_iter = int(sum(p.weight for p in ps))
    burn = list(M.sampler.BURN_IN_FAST) + list(M.sampler.BURN_IN_SLOW)

    def run_device():
        lik = M.SparseLikelihood(prep.lhd, device=dev_index).bind_tree(topo)
        pf = M.PriorFunction(ht, "UncorrelatedLogNormal", cal, [], [], topo, device=dev_index)
        smp = M.Sampler(lik, pf, ps, chains, seed=seed)
        x0 = M.init_with(topo, prep.mean_lengths)
        x0.time_height = ht
        smp.set_initial_state(x0)
        t0 = time.perf_counter()
        for p in burn:                                      # burn-in with the monitors running (the reference's monitor files hold it)
            MO.collect(smp, p, period=period)
            smp.autotune()
        tb = time.perf_counter()
        tr = MO.collect(smp, iterations, period=period)
        t1 = time.perf_counter()
        return smp, tr, tb - t0, t1 - tb

    run_device()                                            # (untimed: code objects, allocations)
    smp, tr, burn_s, run_s = run_device()
    ages = tr.ages()[:, :, post["nodes"]].reshape(-1, len(post["nodes"]))
    ref = {k: np.array(v) for k, v in post["pooled"].items()}
    dev = np.abs(ages.mean(axis=0) - ref["mean"]) / ref["mean"]
    assert np.all(dev <= 0.01), f"node ages off the reference's committed samples: {dev}"
    wall = burn_s + run_s
    n_it = sum(burn) + iterations
    out = {"analysis": "mtCDNApri posterior, SparseMultivariateNormal 0.1, uncorrelated log-normal clock (bench/comparison_with_mcmctree/README.md:615-632)",
           "chains": int(chains), "iterations": int(iterations), "burn_in_iterations": int(sum(burn)), "proposals_per_iteration": S_iter,
           "monitor_period": int(period), "wall_s": wall, "burn_in_s": burn_s, "run_s": run_s, "prepare_s_host": prepare_s,
           "s_per_chain": wall / chains, "proposal_steps_per_s": chains * n_it * S_iter / wall,
           "node_age_max_rel_dev_vs_reference_samples": float(dev.max()), "node_age_rel_dev": [float(x) for x in dev],
           "likelihood": "native sparse handle (mcd_sparse_create / mcd_mh_create_sparse)", "path": smp.last_path(),
           "published_reference": {"value": 154.0, "unit": "s per chain (one chain per process)", "hardware": "Intel i7-1165G7 (Lenovo X1 Carbon Gen 9)",
                                   "source": "bench/comparison_with_mcmctree/README.md:720; scripts/Benchmarking_comptime.R:39-45",
                                   "note": "other hardware, the Haskell binary: a stated baseline beside this figure, not a measured ratio"},
           "haskell_toolchain": haskell_toolchain()}
    # the CPU twin on this box: a bounded sample of the same analysis (one chain per host thread), extrapolated to the whole analysis
    try:
        import oracle as O

        O.build(native=True, force=True)
        nthr = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
        n = topo.n_nodes - 2
        P = np.zeros((n, n))
        for (i, j), v in prep.lhd.sigma_inv_assoc:
            P[i, j] = v
        spec = O.PriorSpec(topo.parent, ht, "UncorrelatedLogNormal", [(c.node, c.lower, c.lower_p, c.upper, c.upper_p) for c in cal], [], [])
        model = O.MhModel(topo.parent, np.asarray(prep.mu, float), P, float(prep.lhd.logdet_sigma), spec, M.table_arrays(ps))
        x0 = M.init_with(topo, prep.mean_lengths)
        x0.time_height = ht
        s0 = M.StateBatch.from_states([x0] * nthr)
        tw = O.MhChains(model, s0.time_birth_rate, s0.time_death_rate, s0.time_height, s0.heights, s0.rate_mean, s0.rate_variance, s0.rates, seed=seed)
        rng = np.random.default_rng(5)
        tw.run(M.cycle_schedule(ps, 2, rng))
        t0 = time.perf_counter()
        its = 0
        while time.perf_counter() - t0 < cpu_budget_s:
            tw.run(M.cycle_schedule(ps, 20, rng))
            its += 20
        dt = time.perf_counter() - t0
        out["cpu_twin"] = {"kind": "port", "cores": nthr, "chains": nthr, "sample": f"{its} iterations of {nthr} chains (one per host thread) in {dt:.1f} s",
                           "s_per_chain_full_analysis_extrapolated": dt / its * n_it, "proposal_steps_per_s": nthr * its * S_iter / dt,
                           "what": "oracle/mh_oracle.c: sequential restatement of the same proposal cycle, prior and dense-matrix likelihood (not the Haskell binary)"}
    except Exception as e:                                   # (a secondary field must not take the line down)
        out["cpu_twin"] = {"error": repr(e)}
    return out


def self_launch(n_ranks, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks ourselves -- fresh interpreter
    processes created BEFORE this process has made any GPU call (it never makes one: it only waits), one per device, rendezvous
    through the environment exactly as torch.distributed.run sets it (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).
    Rank 0's stdout is ours, so the ONE JSON line comes out where the driver reads it.  On a box with fewer devices than ranks the
    children run as a REHEARSAL (MCD_BENCH_REHEARSAL=1: every rank on cuda:0, control collectives and the swap all-gather through
    gloo, because RCCL refuses two ranks on one device) and the line says so (`"rehearsal": true`, `"devices": 1`).
    The device count is taken in a child process as well, so that this one stays clear of the GPU whatever the runtime does."""
    import socket
    import subprocess

    ndev = 0
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
        ndev = int(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 and out.stdout.strip() else 0
    except (OSError, ValueError, subprocess.TimeoutExpired):
        ndev = 0
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    rehearsal = ndev < n_ranks or os.environ.get("MCD_BENCH_REHEARSAL") == "1"
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   MCD_BENCH_SELF_LAUNCHED="1", MCD_BENCH_DEVICES=str(ndev))
        if rehearsal:
            env["MCD_BENCH_REHEARSAL"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + float(os.environ.get("MCD_BENCH_LAUNCH_TIMEOUT", "3000"))
    alive = list(procs)
    while alive:
        for p in list(alive):
            c = p.poll()
            if c is not None:
                alive.remove(p)
                if c != 0:
                    rc = rc or c
        if rc != 0 or time.time() > deadline:                   # one rank failed (or the run hangs): end exactly the processes started here
            for p in alive:
                p.terminate()
            for p in alive:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    return rc


def rank_report(dist, world, rank, elapsed, K, ctl_dev, rehearsal, comm_ranks=None):
    """What the line says about the ranks of a multi-rank run: every rank's own ms_per_step (all-gathered), the rank count the
    process group reports (and, with --kind mh and a swap phase, the one the C ABI's RCCL communicator reports: ncclCommCount),
    how many devices the box has and whether the run was a one-device rehearsal."""
    import torch

    per = [1e3 * elapsed / K]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl_dev)
        g = torch.empty(world, dtype=torch.float64, device=ctl_dev)
        dist.all_gather_into_tensor(g, t)
        per = [1e3 * float(x) / K for x in g.cpu().tolist()]
    ndev = int(os.environ.get("MCD_BENCH_DEVICES", "0")) or torch.cuda.device_count()
    return {"world_size": int(dist.get_world_size()) if world > 1 else 1, "backend": (dist.get_backend() if world > 1 else None),
            "rccl_comm_ranks": comm_ranks, "ms_per_step_per_rank": per, "devices": ndev, "rehearsal": bool(rehearsal),
            "launched_by": ("bench.py (self-launched child processes)" if os.environ.get("MCD_BENCH_SELF_LAUNCHED") == "1"
                            else "torch.distributed.run" if world > 1 else "single process")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--n", "--dim", dest="n", type=int, default=256, help="MVN dimension (--dim: the spelling that survives torch.distributed.run's "
                    "own option parser, where --n is an ambiguous prefix)")
    ap.add_argument("--chains", type=int, default=512, help="chains per GPU")
    ap.add_argument("--swap-period", type=int, default=0, help="--kind mh: MC3 swap phase (all-gather of the ln posteriors + swaps) every P "
                    "iterations of the proposal cycle, config 5; other kinds: all-gather ll every P steps (0 = off)")
    ap.add_argument("--tune-periods", type=int, default=0, help="--kind mh: auto-tuning periods (one iteration of the cycle each) before the warm-up: "
                    "the reference samples with tuned proposals; default 0 = the initial tuning parameters (most proposals accepted)")
    ap.add_argument("--swap-steps", type=int, default=0, help="--kind mh: swap phase every Q lock steps instead (rehearsals shorter than an iteration)")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of hipGraph replay")
    ap.add_argument("--graph-chunk", type=int, default=100, help="steps captured per hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--form", default="auto", choices=["auto", "sweep", "multiply"],
                    help="log-density kernel form (mcd_set_logpdf_form); auto = multiply for N >= 96 and >= 2048 chains, N >= 32 and >= 8192")
    ap.add_argument("--kind", default="logpdf", choices=["logpdf", "grad", "tree", "tree_grad", "prior", "posterior", "mh", "sparse", "e2e"])
    ap.add_argument("--no-mh", action="store_true", help="skip the secondary Metropolis-Hastings measurement of the default run")
    ap.add_argument("--sparse", action="store_true", help="--kind mh: the likelihood over a sparse precision matrix (mcd_mh_create_sparse), the reference's "
                    "production configuration; any --dim up to 2046")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare (`python bench.py --gpus N`): this process becomes the launcher and never touches the GPU
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    import mcmc_date_amd as M
    from mcmc_date_amd import synthetic as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Rehearsal on a one-GPU box: MCD_BENCH_REHEARSAL=1 maps every rank to cuda:0 and uses gloo for the control
    # collectives (RCCL refuses two ranks on one device).  Never set by the driver.
    rehearsal = os.environ.get("MCD_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ctl_dev = torch.device("cpu") if rehearsal else dev   # where the control tensors of the collectives live

    n, B = args.n, args.chains
    if args.kind == "mh":
        # the metric's second reading, as the headline of this run: Metropolis-Hastings proposal steps/s x chains
        K, W = args.steps, args.warmup
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        r = mh_measure(dev_index, n, B, K, W, seed=3, rank=rank, world=world, swap_period=args.swap_period, swap_steps=args.swap_steps, tune_periods=args.tune_periods,
                       rehearsal=rehearsal, sparse=args.sparse)
        elapsed = K * r["us_per_lockstep"] * 1e-6
        ranks = rank_report(dist, world, rank, elapsed, K, ctl_dev, rehearsal, r.get("rccl_comm_ranks"))
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=ctl_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        if rank == 0:
            nd = r["dimension"]
            mh_traffic, mh_traffic_source = None, None
            try:                                             # measured once per round by tools/collect_profiles_r0N.sh
                import glob
                fn = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]
                tr = json.load(open(fn))
                key = {(257, 512, False): "mh_257x512", (1025, 512, False): "mh_1025x512_segments", (2013, 512, True): "mh_sparse_2013x512",
                       (1025, 512, True): "mh_sparse_1025x512"}.get((r["n_nodes"], B, args.sparse))
                if key and key in tr:
                    mh_traffic = tr[key]["fetch_bytes_per_lock_step_corrected"] + tr[key].get("write_bytes_per_lock_step", 0.0)
                    mh_traffic_source = f"profiles/{os.path.basename(fn)}[{key}]"
            except (OSError, ValueError, KeyError, IndexError):
                pass
            alg_b = (8.0 * (2 * r["n_nodes"] + 2) + 8.0 + (8.0 * nd + 4.0 * nd * (nd + 1)) / B) * B   # SURVEY.md 8(d), tree-state kernel
            print(json.dumps({
                "metric": "MVN log-likelihood evals/sec (= MCMC steps/sec \u00d7 chains) at N=256 nodes",
                "value": B * K * world / elapsed, "unit": "MH proposal steps/s (lock steps x chains)", "n_gpus": world, "steps": K, "warmup": W,
                "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic",
                "config": {"workload": f"lock-step Metropolis-Hastings, synthetic {r['n_nodes']}-node tree (dimension {nd}), {B} chains per GPU, "
                                       f"the reference's proposal cycle ({r['proposals_per_iteration']} proposals per iteration)",
                           "n": nd, "chains_per_gpu": B, "kernel": "mh", "launch": "see mh.what",
                           "swap_period": args.swap_period,
                           "parallelism": (f"chains sharded x{world}, no data-path collective" if "mc3" not in r else
                                           f"chains sharded x{world}; MC3 swap phase every {r['mc3']['period_lock_steps']} lock steps: one all-gather of "
                                           f"{r['mc3']['bytes_gathered_per_phase']} bytes + swaps of temperatures")},
                "roofline": {"bound": "hbm", "achieved": alg_b / (elapsed / K) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": alg_b / (elapsed / K) / 1e9 / HBM_PEAK_GBS, "traffic": mh_traffic, "traffic_source": mh_traffic_source,
                             "note": "achieved: SURVEY.md 8(d)'s bytes of ONE FULL likelihood evaluation of the batch per lock step; most proposals are "
                                     "evaluated incrementally (columns of L^-1 on a kept z) and move less -- traffic: memory-side bytes per lock step of "
                                     "the whole run, every kernel (PMC, profiles/)"},
                "ranks": ranks, "mh": r}))
        if world > 1:
            dist.destroy_process_group()
        return
    if args.kind == "e2e":
        if world > 1:
            raise SystemExit("bench.py --kind e2e runs on one GPU (the analysis is one batch of chains)")
        chains = args.chains if args.chains != 512 else 128
        r = e2e_measure(dev_index, chains=chains)
        n_it = r["iterations"] + r["burn_in_iterations"]
        print(json.dumps({
            "metric": "MVN log-likelihood evals/sec (= MCMC steps/sec \u00d7 chains) at N=256 nodes", "value": r["proposal_steps_per_s"],
            "unit": "MH proposal steps/s (lock steps x chains), whole analysis incl. burn-in, tuning and period-2 state fetches",
            "n_gpus": 1, "steps": n_it * r["proposals_per_iteration"], "warmup": 0, "ms_per_step": 1e3 * r["wall_s"] / (n_it * r["proposals_per_iteration"]),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "the reference's committed mtCDNApri inputs (tests/golden)",
            "config": {"workload": f"end to end: mtCDNApri posterior analysis (7 taxa, N = 11, sparse likelihood), {chains} chains, burn-in + 8000 iterations, "
                                   "monitors every 2 iterations", "n": 11, "chains_per_gpu": chains, "kernel": "e2e"},
            "roofline": None, "e2e": r, "cpu_baseline": {**({k: v for k, v in r["cpu_twin"].items()} if "error" not in r["cpu_twin"] else {}),
                                                           "value": r["cpu_twin"].get("proposal_steps_per_s"), "unit": "MH proposal steps/s"}}))
        return
    if args.kind == "sparse":
        K, W = args.steps, args.warmup
        dt, per, r = sparse_measure(dev_index, n, B, K, W)
        if rank == 0:
            r.pop("chains_per_tile")
            print(json.dumps({
                "metric": "MVN log-likelihood evals/sec (= MCMC steps/sec \u00d7 chains) at N=256 nodes", "value": B * K / dt, "unit": "evals/s",
                "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"sparse form: synthetic {n}-dimensional MVN, precision matrix with {r['nnz']} nonzeros (band 3 + 4 random per row), "
                                       f"{B} chains per GPU", "n": n, "chains_per_gpu": B, "kernel": "sparse", "launch": "eager"},
                "roofline": r.pop("roofline"), "sparse": r}))
        if world > 1:
            dist.destroy_process_group()
        return
    if args.kind in ("tree", "tree_grad", "prior", "posterior"):
        topo = S.random_topology((n + 3) // 2, seed=n)
        n = topo.n_nodes - 2
    mu, sigma = S.random_spd_problem(n, seed=n)
    lik = M.MvnLikelihood.from_covariance(mu, sigma, device=dev_index)
    X_host = S.sample_chains(mu, sigma, B, seed=n + 1000 * rank)
    X = torch.as_tensor(X_host, device=dev)
    ll = torch.empty(B, dtype=torch.float64, device=dev)
    lib = M._capi.lib()
    if args.form != "auto":
        M.set_logpdf_form(args.form)

    if args.kind == "logpdf":
        # two input batches, alternated launch by launch, each with its own output: a launch that picked up anything left
        # behind by its predecessor (the row-split form hands partial sums over through a scratch) cannot pass for correct
        X2 = torch.as_tensor(S.sample_chains(mu, sigma, B, seed=n + 1000 * rank + 500), device=dev)
        ll2 = torch.empty(B, dtype=torch.float64, device=dev)
        flip = [0]

        def step():
            if flip[0] == 0:
                lik.logpdf_into(X, ll)
            else:
                lik.logpdf_into(X2, ll2)
            flip[0] ^= 1
    elif args.kind == "grad":
        G = torch.empty_like(X)

        def step():
            M._capi.check(lib.mcd_mvn_grad_batch(lik._h, X.data_ptr(), X.stride(0), B, 1,
                                                 torch.cuda.current_stream().cuda_stream, ll.data_ptr(), G.data_ptr(), G.stride(0)))
    else:
        tl = lik.bind_tree(topo)
        st = S.random_states(topo, B, seed=n + 1000 * rank).to(dev)
        lj = torch.empty(B, dtype=torch.float64, device=dev)
        gH, gR = torch.empty_like(st.heights), torch.empty_like(st.rates)
        gt, gm = torch.empty_like(st.time_height), torch.empty_like(st.rate_mean)
        if args.kind in ("prior", "posterior"):
            rng = np.random.default_rng(n)
            st.time_birth_rate = torch.as_tensor(np.exp(0.3 * rng.standard_normal(B)), device=dev)
            st.time_death_rate = torch.as_tensor(np.exp(0.3 * rng.standard_normal(B)), device=dev)
            st.rate_variance = torch.as_tensor(0.2 + rng.random(B), device=dev)
            pf = M.PriorFunction(1.0, "UncorrelatedLogNormal", [M.Calibration("root", 0, 0.9, 0.025, 1.1, 0.025)],
                                 [M.Constraint("k", 7, 3, 0.025)], [], topo, device=dev_index)
            lp = torch.empty(B, dtype=torch.float64, device=dev)

            def prior_step():
                M._capi.check(lib.mcd_prior_logprior_batch(pf._p, st.time_birth_rate.data_ptr(), st.time_death_rate.data_ptr(),
                                                           st.time_height.data_ptr(), st.heights.data_ptr(), st.rate_mean.data_ptr(),
                                                           st.rate_variance.data_ptr(), st.rates.data_ptr(), st.heights.stride(0), B, 1,
                                                           torch.cuda.current_stream().cuda_stream, lp.data_ptr(), None))

            def lik_step():
                M._capi.check(lib.mcd_tree_loglik_batch(tl._t, st.heights.data_ptr(), st.rates.data_ptr(), st.heights.stride(0),
                                                        st.time_height.data_ptr(), st.rate_mean.data_ptr(), B, 1,
                                                        torch.cuda.current_stream().cuda_stream, ll.data_ptr(), lj.data_ptr()))

            if args.kind == "prior":
                step = prior_step
            else:
                def step():
                    prior_step()
                    lik_step()
        elif args.kind == "tree":
            def step():
                M._capi.check(lib.mcd_tree_loglik_batch(tl._t, st.heights.data_ptr(), st.rates.data_ptr(), st.heights.stride(0),
                                                        st.time_height.data_ptr(), st.rate_mean.data_ptr(), B, 1,
                                                        torch.cuda.current_stream().cuda_stream, ll.data_ptr(), lj.data_ptr()))
        else:
            def step():
                M._capi.check(lib.mcd_tree_grad_batch(tl._t, st.heights.data_ptr(), st.rates.data_ptr(), st.heights.stride(0),
                                                      st.time_height.data_ptr(), st.rate_mean.data_ptr(), B, 1,
                                                      torch.cuda.current_stream().cuda_stream, ll.data_ptr(), gH.data_ptr(),
                                                      gR.data_ptr(), gt.data_ptr(), gm.data_ptr()))

    gathered = torch.empty(world * B, dtype=torch.float64, device=ctl_dev) if (world > 1 and args.swap_period > 0) else None
    # which form the launch takes (mirror of use_wide / use_wide_grad in csrc/k_logpdf.hip)
    has_grad = args.kind in ("grad", "tree_grad")
    if args.kind == "prior":
        form = "sweep"
    elif args.form != "auto":
        form = args.form
    else:
        form = "multiply" if ((n >= 96 and B >= 2048) or (n >= 32 and B >= 8192)) else "sweep"
        if (args.kind in ("logpdf", "tree") and B <= 1024 and (n > 256 or (n > 240 and B <= 128))) or \
                (has_grad and B <= 1024 and (n > 256 or (n > 240 and B <= 512))):
            form = "split"                               # k_split.hip: W's row blocks over 8-32 workgroups per chain tile
    # hipGraph replay hides the per-launch dispatch cost of the few-microsecond sweep launches; the multiply form's launches
    # are longer than an eager dispatch and are launched eagerly.  (On this pool about one run in twenty sees the host learn
    # of the completion ~60 ms late -- graph or eager, blocking wait or polling, HSA_ENABLE_INTERRUPT=0 or not -- while the HIP
    # events of the same run agree with the kernel trace; the default run is long enough that such an outlier costs < 10 %.)
    use_graph = (not args.no_graph) and gathered is None and form in ("sweep", "split")
    K, W = args.steps, args.warmup

    # --- build the launch schedule -----------------------------------------------------------
    graphs = []
    if use_graph:
        chunk = max(1, min(args.graph_chunk, K))
        sizes = sorted({chunk, K % chunk} - {0})
        cap_stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(cap_stream):
            step()                                       # load the code object before capture
        cap_stream.synchronize()
        gmap = {}
        for sz in sizes:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap_stream):
                for _ in range(sz):
                    step()
            gmap[sz] = g
        for g in gmap.values():                          # the first replay uploads the graph: keep that out of the timed
            g.replay()                                   # region whatever --warmup is
        torch.cuda.synchronize()

        def run(k):
            full, rem = divmod(k, chunk)
            for _ in range(full):
                gmap[chunk].replay()
            if rem:
                if rem in gmap:
                    gmap[rem].replay()
                else:
                    for _ in range(rem):
                        step()
    else:
        def run(k):
            for i in range(k):
                step()
                if gathered is not None and (i + 1) % args.swap_period == 0:
                    if rehearsal:
                        dist.all_gather_into_tensor(gathered, ll.cpu())
                    else:
                        dist.all_gather_into_tensor(gathered, ll)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up = W untimed steps; the last of them go through the same sequence as the timed region (timing events,
    # replay, polling) so that nothing in it runs for the first time under the clock
    r = min(W, args.graph_chunk) if use_graph else min(W, 10)
    run(W - r)
    fence()
    if r > 0:
        re0, re1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        re0.record()
        run(r)
        re1.record()
        while not re1.query():
            pass
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(K)
    ev1.record()
    while not ev1.query():                               # poll rather than block in the driver
        pass
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    dev_ms = ev0.elapsed_time(ev1)
    ranks = rank_report(dist, world, rank, elapsed, K, ctl_dev, rehearsal)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the results of the last steps are the oracle-checked quantity (finite; for the log-density, both alternating
    # batches against a separate evaluation by the column sweep on another handle)
    ll_host = ll.cpu().numpy()
    assert np.all(np.isfinite(ll_host)) or os.environ.get("MCD_SPLIT_PROBE"), "non-finite log-likelihood in the bench batch"
    if args.kind == "logpdf" and not os.environ.get("MCD_SPLIT_PROBE"):
        chk = M.MvnLikelihood.from_covariance(mu, sigma, device=dev_index)
        chk.set_form("sweep")
        for xs, out in ((X, ll), (X2, ll2)):
            ref = chk.logpdf(xs)
            err = float(((out - ref).abs() / ref.abs()).max())
            assert err <= 1e-11, f"bench batch differs from the sweep form: {err}"

    if rank == 0:
        evals = float(K) * B * world
        per_launch_s = (dev_ms * 1e-3) / K
        alg_b = algorithmic_bytes_per_eval(n, B) * B
        achieved = alg_b / per_launch_s / 1e9
        flops = algorithmic_flops_per_eval(n) * (2.0 if has_grad else 1.0) * B / per_launch_s / 1e12
        traffic, traffic_source = (measured_traffic({("logpdf", 256): "n256", ("logpdf", 1024): "n1024", ("tree", 255): "tree255",
                                                     ("tree", 1023): "tree1023"}.get((args.kind, n)))
                                   if B == 512 and args.form == "auto" and world == 1 else (None, None))
        out = {
            "metric": "MVN log-likelihood evals/sec (= MCMC steps/sec \u00d7 chains) at N=256 nodes",   # BASELINE.json:metric, verbatim
            "value": evals / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"synthetic {n}-dimensional MVN (dense random SPD Sigma, seed {n}), {B} chains per GPU, "
                                   f"kind={args.kind}", "n": n, "chains_per_gpu": B, "kernel": args.kind,
                       "launch": "hipGraph replay" if use_graph else "eager",
                       "swap_period": args.swap_period, "parallelism": (f"chains sharded x{world}, no data-path collective" if gathered is None
                                       else f"chains sharded x{world} + ll all-gather every {args.swap_period} steps")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_us_per_launch": per_launch_s * 1e6,
                         "alg_bytes_per_launch": alg_b,
                         "fp64_tflops": flops, "fp64_frac": flops / FP64_PEAK_TFLOPS},
        }
        out["config"]["form"] = form
        if world > 1:
            out["ranks"] = ranks
        if form == "multiply":
            # k_wide.hip: a triangular matrix product on the fp64 matrix cores -- priced against the dense fp64 MFMA peak
            # (v_mfma_f64_16x16x4_f64: 64 cycles per 16x16x4 tile product and SIMD = the vector fp64 rate, 78.6 TFLOP/s)
            out["roofline"].update({"bound": "mfma", "achieved": flops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": flops / FP64_PEAK_TFLOPS, "hbm_gbs": achieved})
            if n == 256 and B == 8192 and args.kind == "logpdf":     # the configuration the committed PMC passes were run on
                try:
                    import glob
                    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_wide_summary.json")))
                    out["roofline"]["traffic"] = float(json.load(open(fs[-1]))["pmc_traffic_n256_b8192"]["per_launch_bytes_corrected"])
                    out["roofline"]["traffic_source"] = "profiles/" + os.path.basename(fs[-1]) + "[pmc_traffic_n256_b8192] (committed rocprofv3 --pmc passes)"
                except Exception:
                    pass
        if world == 1 and args.kind == "logpdf" and not args.no_mh:
            # the metric's "= MCMC steps/sec x chains": real Metropolis-Hastings steps on a tree of this size (secondary field)
            out["mh"] = mh_measure(dev_index, n, B, 4000, 400, repeats=2)
            if n == 256 and B == 512:
                # ... and on BASELINE config 5's share of one GPU (1025-node tree, 512 chains; `--kind mh --dim 1024` is the full line)
                try:
                    r5 = mh_measure(dev_index, 1024, 512, 4000, 400, repeats=2)
                    out["mh_config5_share"] = {k: r5[k] for k in ("value", "unit", "us_per_lockstep", "n_nodes", "dimension", "chains", "lock_steps", "lds_bytes_per_workgroup", "acceptance_rate", "what")}
                except Exception as e:                       # (a secondary field must not take the line down)
                    out["mh_config5_share"] = {"error": repr(e)}
            if n == 256 and B == 512:
                # ... and over a SPARSE likelihood, the reference's production configuration (every published timing of the reference uses it): at
                # config 5's size and at the size of its 1007-taxon example
                try:
                    out["mh_sparse"] = []
                    for dim in (1024, 2012):
                        rs = mh_measure(dev_index, dim, 512, 4000, 400, sparse=True, repeats=2)
                        out["mh_sparse"].append({k: rs[k] for k in ("value", "unit", "us_per_lockstep", "n_nodes", "dimension", "chains", "lock_steps", "likelihood", "lds_bytes_per_workgroup", "acceptance_rate", "what")})
                except Exception as e:
                    out["mh_sparse"] = {"error": repr(e)}
        if world == 1 and args.kind == "logpdf" and not args.no_mh and B <= 1024:
            out["full_gpu"] = full_gpu_measure(dev_index, n)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, mu, sigma, X_host)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
