"""Multi-process CPU test (gloo, world_size 2) of the N > 1 path: chain sharding + the swap/diagnostic
all-gather of per-chain log-likelihoods (BASELINE.json config 5).  The likelihood values themselves come
from the CPU oracle here (the HIP path needs a GPU); what is under test is the sharding arithmetic, the
global ordering of the gathered vector and the ragged-shard case."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_chains, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    from mcmc_date_amd import synthetic as S
    from mcmc_date_amd.shards import ChainShard, gather_loglik

    n = 24
    mu, sigma = S.random_spd_problem(n, seed=5)
    X = S.sample_chains(mu, sigma, n_chains, seed=5)            # every rank can generate the global batch
    P = np.linalg.inv(sigma)
    logdet = np.linalg.slogdet(sigma)[1]
    sh = ChainShard(rank, world, n_chains)
    ll_local = torch.as_tensor(O.logpdf_full_batch(mu, P, logdet, X[sh.lo:sh.hi]))
    for step in range(4):                                       # all-gather every 2 steps (SwapPeriod 2)
        if (step + 1) % 2 == 0:
            g = gather_loglik(ll_local, sh)
    full = O.logpdf_full_batch(mu, P, logdet, X)
    q.put((rank, bool(np.array_equal(g.numpy(), full)), sh.lo, sh.hi))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_chains", [64, 37])
def test_sharded_chains_gather_in_global_order(n_chains):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_chains, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=180) for _ in range(world))
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    assert all(ok for _, ok, _, _ in res)
    assert res[0][2] == 0 and res[0][3] == res[1][2] and res[1][3] == n_chains   # contiguous cover


def test_shard_arithmetic():
    from mcmc_date_amd.shards import ChainShard

    for n, w in [(4096, 8), (37, 2), (5, 8), (512, 1)]:
        shards = [ChainShard(r, w, n) for r in range(w)]
        assert shards[0].lo == 0 and shards[-1].hi == n
        assert all(a.hi == b.lo for a, b in zip(shards, shards[1:]))
        assert sum(s.size for s in shards) == n and max(s.size for s in shards) - min(s.size for s in shards) <= 1


def _mh_worker(rank, world, port, n_chains, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mcmc_date_amd as M
    import oracle as O
    from mcmc_date_amd.shards import ChainShard, gather_chain_rows

    fx = dict(np.load(os.path.join(ROOT, "tests", "golden", "06-leaves-constant-rate.npz")))
    topo = M.Topology(fx["parent"])
    ps, _ = M.proposals(topo, [], calibrations_available=False)
    spec = O.PriorSpec(fx["parent"], float(fx["prior_ht"]), "UncorrelatedGamma", [], [], [])
    model = O.MhModel(fx["parent"], fx["mu"], fx["sigma_inv"], float(fx["logdet"]), spec, M.table_arrays(ps))
    x0 = M.init_with(topo, fx["mean_lengths"])
    sched = M.cycle_schedule(ps, 6, np.random.default_rng(123))          # the schedule depends on the seed only: same on every rank

    def chains(lo, hi):
        s = M.StateBatch.from_states([x0] * (hi - lo))
        return O.MhChains(model, s.time_birth_rate, s.time_death_rate, s.time_height, s.heights, s.rate_mean, s.rate_variance, s.rates,
                          seed=9, chain0=lo)

    sh = ChainShard(rank, world, n_chains)
    mine = chains(sh.lo, sh.hi)
    mine.run(sched, accumulate=True)
    ages = gather_chain_rows(torch.as_tensor(mine.age_sum / mine.n_samples), sh)
    post = gather_chain_rows(torch.as_tensor(mine.post), sh)
    full = chains(0, n_chains)
    full.run(sched, accumulate=True)
    ok = np.array_equal(ages.numpy(), full.age_sum / full.n_samples) and np.array_equal(post.numpy(), full.post)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_chains", [8, 5])
def test_sharded_mh_chains_equal_the_unsharded_run(n_chains):
    """N > 1 path of the sampler: every rank advances its block of chains with the global chain index as the random
    stream id (what shards.shard_sampler passes as first_chain on a GPU); the gathered per-chain results equal a
    single-process run bit for bit.  The chains here are the CPU twin (the HIP path needs a GPU)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_mh_worker, args=(r, world, port, n_chains, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=240) for _ in range(world))
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps) and all(ok for _, ok in res)


def _mc3_worker(rank, world, port, n_chains, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mcmc_date_amd as M
    import oracle as O
    from mcmc_date_amd.shards import ChainShard, gather_chain_rows, gather_posterior_host
    from twin_backend import TwinBackend

    fx = dict(np.load(os.path.join(ROOT, "tests", "golden", "06-leaves-constant-rate.npz")))
    topo = M.Topology(fx["parent"])
    ps, _ = M.proposals(topo, [], calibrations_available=False)
    spec = O.PriorSpec(fx["parent"], float(fx["prior_ht"]), "UncorrelatedGamma", [], [], [])
    model = O.MhModel(fx["parent"], fx["mu"], fx["sigma_inv"], float(fx["logdet"]), spec, M.table_arrays(ps))
    x0 = M.init_with(topo, fx["mean_lengths"])

    def backend(lo, hi):
        s = M.StateBatch.from_states([x0] * (hi - lo))
        return TwinBackend(O.MhChains(model, s.time_birth_rate, s.time_death_rate, s.time_height, s.heights, s.rate_mean, s.rate_variance,
                                      s.rates, seed=9, chain0=lo), ps, seed=9)

    ladder = [1.0, 0.8, 0.6, 0.4]                                          # a steep ladder: swaps are accepted and refused
    sh = ChainShard(rank, world, n_chains)
    mine = backend(sh.lo, sh.hi)
    mc3 = M.MC3(mine, n_chains=4, swap_period=2, n_swaps=3, betas=ladder, seed=5, shard=sh,
                gather=lambda local: gather_posterior_host(local, sh))
    mc3.run(12)
    full = backend(0, n_chains)
    ref = M.MC3(full, n_chains=4, swap_period=2, n_swaps=3, betas=ladder, seed=5)
    ref.run(12)
    st, sf = mine.state(), full.state()
    ok = (np.array_equal(mc3.rank, ref.rank) and np.array_equal(mc3.swaps_accepted, ref.swaps_accepted)
          and np.array_equal(st.heights, sf.heights[sh.lo:sh.hi]) and np.array_equal(st.rates, sf.rates[sh.lo:sh.hi])
          and np.array_equal(mine.c.beta, full.c.beta[sh.lo:sh.hi]) and np.array_equal(mine.c.post, full.c.post[sh.lo:sh.hi]))
    moved = bool(ref.swaps_accepted.sum() > 0 and (ref.swaps_tried - ref.swaps_accepted).sum() > 0 and not np.array_equal(ref.rank, np.arange(n_chains) % 4))
    # the cold chains, gathered: the same rows whatever the number of ranks
    cold_local = np.zeros((sh.size, 1))
    cold_local[mc3.cold(), 0] = 1.0
    cold = gather_chain_rows(torch.as_tensor(cold_local), sh).numpy()[:, 0]
    ok = ok and np.array_equal(np.nonzero(cold)[0], ref.cold()) and len(ref.cold()) == n_chains // 4
    q.put((rank, bool(ok), moved))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_mc3_does_not_depend_on_the_rank_count():
    """BASELINE.json config 5's loop as a rehearsal on the CPU: chains sharded over two ranks, every SwapPeriod = 2 iterations one
    all-gather of the per-chain ln posteriors and a swap phase of NSwaps = 3 adjacent pairs per group of NChains = 4
    (app/Main.hs:476-478).  12 chains over 2 ranks: the middle group (chains 4 .. 7) straddles the ranks, so its swaps need the
    gathered values.  Every rank evaluates all groups on counter-based draws: temperature ranks, swap counters, states and
    cold-chain rows equal the single-process run bit for bit.  (The chains are the CPU twin; on a GPU the same loop runs through
    mcd_mh_run -> mcd_shard_allgather -> mcd_mh_mc3_swap on the sampler's stream: shards.mc3_for_shard.)"""
    world, n_chains = 2, 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_mc3_worker, args=(r, world, port, n_chains, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=240) for _ in range(world))
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps) and all(ok for _, ok, _ in res) and all(moved for _, _, moved in res)
