"""Phase stamps of the row-split kernel (k_split.hip): the waves of group 0 and of group G - 1 of tile 0, cycles (s_memtime)
between: start | staging loads done + LDS written | barrier | MFMA runs | cut blocks + partial sums | store + counter add
returned | (last arriver only) slots read, ll written.
Build the stamped library first:  make -C mcmc-date_amd/csrc stamp_split   (writes tools/microbench/libsplitstamp.so),
then on the GPU box:  python tools/microbench/split_stamps.py [n] [chains] [tree]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MCD_LIB_PATH"] = os.path.join(ROOT, "tools", "microbench", "libsplitstamp.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mcmc_date_amd as M
from mcmc_date_amd import synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
tree = len(sys.argv) > 3
dev = "cuda:0"
if tree:
    topo = S.random_topology((n + 3) // 2, seed=n)
    n = topo.n_nodes - 2
mu, sigma = S.random_spd_problem(n, seed=n)
lik = M.MvnLikelihood.from_covariance(mu, sigma)
X = torch.as_tensor(S.sample_chains(mu, sigma, B, seed=1), device=dev)
out = torch.empty(B, dtype=torch.float64, device=dev)
if tree:
    tl = lik.bind_tree(topo)
    st_ = S.random_states(topo, B, seed=1).to(dev)
L = M._capi.lib()
L.mcd_split_debug_stamps.argtypes = [ctypes.c_void_p]
acc = np.zeros((16, 6))
cnt = np.zeros((16, 6))
reps = 200
for it in range(reps + 5):
    for _ in range(3):                                   # back to back, as a sampler launches them
        if tree:
            tl.loglik(st_)
        else:
            lik.logpdf_into(X, out)
    torch.cuda.synchronize()
    st = np.zeros(256, dtype=np.uint64)
    L.mcd_split_debug_stamps(st.ctypes.data)
    st = st.reshape(16, 16).astype(np.int64)
    if it >= 5:
        d = np.diff(st[:, :7], axis=1)
        ok = (d > 0) & (d < 10**7)
        acc += np.where(ok, d, 0)
        cnt += ok
acc /= np.maximum(cnt, 1)
fine = st[:, [2, 8, 9, 10, 11, 12, 3, 13, 14, 4, 5, 6]]
print("last launch, fine stamps (cycles since the barrier): B reads issued | group 1 | group 2 | group 3 | runs done | T3 | after barrier (wave 0) | cut blocks done | T4 | T5 | T6")
for i in range(16):
    print("  group %s wave %d: %s" % ("0  " if i < 8 else "G-1", i % 8, "  ".join("%6d" % (v - fine[i, 0]) for v in fine[i, 1:])))
print("cycles per phase [n=%d chains=%d%s]: sched | stage | barrier | mfma | combine | store+add | read+ll" % (n, B, " tree" if tree else ""))
for g in range(2):
    for w in range(8):
        r = acc[g * 8 + w]
        print("group %s wave %d: " % ("0  " if g == 0 else "G-1", w) + "  ".join("%7.0f" % v for v in r) + "   total %7.0f" % r.sum())
