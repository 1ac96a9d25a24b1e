"""Chain sharding across the GPUs of one node (one process per GPU, torch.distributed).

Chains are independent (each evaluation depends only on its own state and the shared immutable
operands), so the likelihood path itself needs NO collective: every rank creates its own
`MvnLikelihood` (operands replicated, <= 16 MB at N = 1024) and evaluates a contiguous block of chains.
The only exchange is the sampler-level one of BASELINE.json config 5: an all-gather of the per-chain
log-likelihoods every `swap_period` steps (MC3 swap / convergence diagnostics; the reference's
in-process analogue is `MC3Settings (NChains 4) (SwapPeriod 2) (NSwaps 3)`, app/Main.hs:477).  It is
a few KB per rank, latency bound: one all-gather (RCCL over xGMI on GPUs, gloo on CPU), no ring of
small sends.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class ChainShard:
    """Contiguous block [lo, hi) of the global chain index range owned by one rank."""
    rank: int
    world: int
    n_chains: int

    @property
    def lo(self) -> int:
        q, r = divmod(self.n_chains, self.world)
        return self.rank * q + min(self.rank, r)

    @property
    def hi(self) -> int:
        q, r = divmod(self.n_chains, self.world)
        return self.lo + q + (1 if self.rank < r else 0)

    @property
    def size(self) -> int:
        return self.hi - self.lo

    def counts(self):
        return [ChainShard(r, self.world, self.n_chains).size for r in range(self.world)]


def gather_loglik(ll_local, shard: ChainShard):
    """All-gather of per-chain log-likelihoods in global chain order (ragged shards allowed).

    ll_local: 1-D torch tensor (CUDA with the nccl/RCCL backend, CPU with gloo) of length shard.size.
    Returns a 1-D tensor of length shard.n_chains on every rank."""
    import torch
    import torch.distributed as dist

    if shard.world == 1:
        return ll_local.clone()
    counts = shard.counts()
    if len(set(counts)) == 1:
        out = torch.empty(shard.n_chains, dtype=ll_local.dtype, device=ll_local.device)
        dist.all_gather_into_tensor(out, ll_local.contiguous())
        return out
    m = max(counts)
    pad = torch.zeros(m, dtype=ll_local.dtype, device=ll_local.device)
    pad[: shard.size] = ll_local
    out = torch.empty(shard.world * m, dtype=ll_local.dtype, device=ll_local.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * m: r * m + c] for r, c in enumerate(counts)])


def gather_chain_rows(rows_local, shard: ChainShard):
    """All-gather of a per-chain table ([shard.size, k]: e.g. the sampler's per-chain node-age means or ln posterior
    triples) in global chain order; ragged shards allowed.  One collective, like gather_loglik."""
    import torch
    import torch.distributed as dist

    if rows_local.dim() != 2 or rows_local.shape[0] != shard.size:
        raise ValueError("gather_chain_rows: expected a [shard.size, k] tensor")
    if shard.world == 1:
        return rows_local.clone()
    k = rows_local.shape[1]
    counts = shard.counts()
    m = max(counts)
    pad = torch.zeros(m, k, dtype=rows_local.dtype, device=rows_local.device)
    pad[: shard.size] = rows_local
    out = torch.empty(shard.world * m, k, dtype=rows_local.dtype, device=rows_local.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * m: r * m + c] for r, c in enumerate(counts)])


def shard_sampler(tree_lik, prior, table, n_chains: int, seed: int, shard: ChainShard):
    """This rank's block of a global set of `n_chains` Metropolis-Hastings chains: a `Sampler` whose chain b draws the
    random stream of global chain shard.lo + b, so the global result does not depend on the number of GPUs."""
    from .sampler import Sampler

    return Sampler(tree_lik, prior, table, shard.size, seed, first_chain=shard.lo)


def gather_posterior_host(local, shard: ChainShard):
    """[3][batch] ln prior / ln likelihood / ln Jacobian of this rank's chains (numpy) -> [world][3][batch] on every rank through
    torch.distributed (gloo on CPU): what mcd_shard_allgather does with the device arrays, for rehearsals without a GPU."""
    import numpy as np
    import torch
    import torch.distributed as dist

    if shard.world == 1:
        return np.asarray(local)[None]
    t = torch.as_tensor(np.ascontiguousarray(local, dtype=np.float64))
    out = torch.empty((shard.world * t.shape[0],) + tuple(t.shape[1:]), dtype=torch.float64)
    dist.all_gather_into_tensor(out, t)
    return out.numpy().reshape((shard.world,) + tuple(t.shape))


def gather_posterior_device(sampler, comm: "ShardComm"):
    """The sampler's device-resident [3][batch] posterior array all-gathered over the ranks on the SAMPLER'S stream (RCCL behind the
    C ABI: mcd_mh_posterior_device + mcd_shard_allgather): a [world, 3, batch] CUDA tensor, valid in stream order -- what
    mcd_mh_mc3_swap takes as `gathered`.  No host synchronisation."""
    import ctypes as C

    import torch

    from . import _capi

    lib = _capi.lib()
    dptr, st = C.c_void_p(), C.c_void_p()
    _capi.check(lib.mcd_mh_posterior_device(sampler._h, C.byref(dptr), C.byref(st)))
    # The receive buffer belongs to the communicator and lives as long as it does: it is written by the all-gather and read by
    # the swap kernel on the SAMPLER'S stream only, which torch's caching allocator knows nothing about -- a fresh torch.empty per
    # phase could be handed out again (or still carry pending work of torch's stream) while that stream uses it.
    key = (int(sampler.batch),)
    out = comm._gathered.get(key)
    if out is None:
        dev = torch.device("cuda", torch.cuda.current_device())
        out = torch.empty((comm.shard.world, 3, sampler.batch), dtype=torch.float64, device=dev)
        torch.cuda.current_stream(dev).synchronize()          # nothing of torch's stream is pending on the block when the other stream gets it
        comm._gathered[key] = out
    _capi.check(lib.mcd_shard_allgather(comm._comm, dptr, C.c_void_p(out.data_ptr()), 3 * sampler.batch, st))
    return out


def mc3_for_shard(sampler, shard: ChainShard, comm=None, **kw):
    """Metropolis-coupled MCMC over a sharded set of chains (BASELINE.json config 5: `mc3 (MC3Settings (NChains 4) (SwapPeriod 2)
    (NSwaps 3))`, app/Main.hs:476-478, over 8 GPUs): `sampler` = shard_sampler(...) of this rank, `comm` = its ShardComm (None on
    one rank).  One period = sampler.run(swap_period) -> all-gather of the [3][batch] ln posteriors on the sampler's stream ->
    mcd_mh_mc3_swap: sampler.MC3 with this rank's gather."""
    from .sampler import MC3

    gather = None if shard.world == 1 else (lambda smp: gather_posterior_device(smp, comm))
    return MC3(sampler, shard=shard, gather=gather, **kw)


class ShardComm:
    """The C ABI's communicator (include/mcmcdate_mvn.h: mcd_shard_*; RCCL's ncclAllGather bound at run time) -- what a host in
    the reference's language would use.  The unique id is drawn by rank 0 and handed to the other ranks through `exchange`, a
    callable bytes -> bytes that broadcasts rank 0's argument (default: torch.distributed's object broadcast when a process
    group exists; with one rank nothing is exchanged)."""

    def __init__(self, shard: ChainShard, device: int = 0, exchange=None):
        import ctypes as C

        from . import _capi

        self.shard = shard
        self._lib = _capi.lib()
        self._gathered = {}
        buf = C.create_string_buffer(128)
        if shard.rank == 0:
            _capi.check(self._lib.mcd_shard_unique_id(buf))
        ident = buf.raw
        if shard.world > 1:
            if exchange is None:
                # a 128-byte tensor broadcast: on the device for the nccl (= RCCL) backend, on the host for gloo -- no pickled
                # objects, nothing that depends on how the backend moves Python objects
                import torch
                import torch.distributed as dist

                on_dev = dist.get_backend() == "nccl"
                t = torch.frombuffer(bytearray(ident), dtype=torch.uint8).clone()
                if on_dev:
                    t = t.to(torch.device("cuda", int(device)))
                dist.broadcast(t, src=0)
                ident = bytes(t.cpu().numpy().tobytes())
            else:
                ident = exchange(ident)
        self._comm = C.c_void_p()
        _capi.check(self._lib.mcd_shard_comm_create(C.byref(self._comm), shard.world, shard.rank, ident, int(device)))

    def count(self) -> int:
        """The rank count the RCCL communicator itself reports (ncclCommCount)."""
        import ctypes as C

        from . import _capi

        n = C.c_int(0)
        _capi.check(self._lib.mcd_shard_comm_count(self._comm, C.byref(n)))
        return int(n.value)

    def allgather(self, send, stream=None):
        """send: 1-D float64 CUDA tensor, the same length on every rank; returns [world * len] in rank order."""
        import ctypes as C

        import torch

        from . import _capi

        send = send.contiguous()
        out = torch.empty(self.shard.world * send.numel(), dtype=torch.float64, device=send.device)
        st = C.c_void_p(torch.cuda.current_stream(send.device).cuda_stream if stream is None else stream)
        _capi.check(self._lib.mcd_shard_allgather(self._comm, C.c_void_p(send.data_ptr()), C.c_void_p(out.data_ptr()), send.numel(), st))
        return out

    def close(self):
        if getattr(self, "_comm", None) is not None and self._comm.value:
            self._lib.mcd_shard_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
