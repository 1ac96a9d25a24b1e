"""The CPU twin (oracle.MhChains) behind the interface sampler.MC3 and the sharded loops expect of a backend: batch, run(k),
posterior(), set_temperatures(beta), state().  The cycle order of every iteration comes from a generator seeded like
Sampler's, so that a twin backend and a device Sampler with the same seed execute the same schedule -- and so do the ranks of a
sharded rehearsal."""
import numpy as np

import mcmc_date_amd as M


class TwinBackend:
    def __init__(self, chains, table, seed):
        self.c = chains
        self.table = list(table)
        self.batch = chains.B
        self._sched_rng = np.random.default_rng([int(seed), 0x5EED])       # as Sampler.__init__

    def run(self, n_iter, accumulate=False):
        self.c.run(M.cycle_schedule(self.table, n_iter, self._sched_rng), accumulate=accumulate)

    def posterior(self):
        return self.c.post.copy()                                          # [batch, 3]

    def set_temperatures(self, beta):
        self.c.beta[:] = beta

    def state(self):
        c = self.c
        return M.StateBatch(c.H.copy(), c.R.copy(), c.tH.copy(), c.rMu.copy(), c.birth.copy(), c.death.copy(), c.rVar.copy())
