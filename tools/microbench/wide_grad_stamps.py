"""Phase stamps of the multiply-form tree gradient (k_wide_grad.hip), workgroup 0, every wave (cycles):
stage | forward | z barrier+store | backward | g barrier+store | chain-rule loads + gR | g.d exchange | e scatter | gH gather.
Build first:  make -C mcmc-date_amd/csrc stamp_wide_grad  (writes tools/microbench/libwidegradstamp.so),
then on the GPU box:  python tools/microbench/wide_grad_stamps.py [n_leaves] [chains]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MCD_LIB_PATH"] = os.path.join(ROOT, "tools", "microbench", "libwidegradstamp.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mcmc_date_amd as M
from mcmc_date_amd import synthetic as S
leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
topo = S.random_topology(leaves, seed=3)
n = topo.n_nodes - 2
mu, sigma = S.random_spd_problem(n, seed=3)
tl = M.MvnLikelihood.from_covariance(mu, sigma).bind_tree(topo)
st = S.random_states(topo, B, seed=4).to("cuda:0")
M.set_logpdf_form("multiply")
L = M._capi.lib()
L.mcd_wide_debug_stamps.argtypes = [ctypes.c_void_p]
acc = np.zeros((8, 9))
reps = 30
for i in range(reps + 5):
    tl.grad(st)
    torch.cuda.synchronize()
    s = np.zeros(128, dtype=np.uint64)
    L.mcd_wide_debug_stamps(s.ctypes.data)
    s = s.reshape(8, 16).astype(np.int64)
    if i >= 5:
        acc += np.diff(s[:, :10], axis=1)
acc /= reps
print("cycles per phase (rows = waves) [n=%d chains=%d]" % (n, B))
print("         stage  forward  z-store  backward g-store  loads+gR  g.d-xchg  e-scatter  gH-gather")
for w in range(8):
    print("wave %d: " % w + " ".join("%8.0f" % v for v in acc[w]) + "   total %8.0f" % acc[w].sum())
