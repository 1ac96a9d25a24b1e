"""Phase stamps of the multiply-form kernel (k_wide.hip), workgroup 0, every wave: cycles (s_memtime, 100 MHz ticks
converted) spent before the first barrier, staging, to the second barrier, in the MFMA loop, in the reduction.
Build the stamped library first:  make -C mcmc-date_amd/csrc stamp_wide   (writes tools/microbench/libwidestamp.so),
then on the GPU box:  python tools/microbench/wide_stamps.py [n] [chains]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MCD_LIB_PATH"] = os.path.join(ROOT, "tools", "microbench", "libwidestamp.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mcmc_date_amd as M
from mcmc_date_amd import synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
mu, sigma = S.random_spd_problem(n, seed=n)
lik = M.MvnLikelihood.from_covariance(mu, sigma)
X = torch.as_tensor(S.sample_chains(mu, sigma, B, seed=1), device="cuda:0")
out = torch.empty(B, dtype=torch.float64, device="cuda:0")
M.set_logpdf_form("multiply")
L = M._capi.lib()
L.mcd_wide_debug_stamps.argtypes = [ctypes.c_void_p]
acc = np.zeros((8, 5))
reps = 50
for _ in range(reps + 5):
    lik.logpdf_into(X, out)
    torch.cuda.synchronize()
    st = np.zeros(128, dtype=np.uint64)
    L.mcd_wide_debug_stamps(st.ctypes.data)
    st = st.reshape(8, 16).astype(np.int64)
    if _ >= 5:
        acc += np.diff(st[:, :6], axis=1)
acc /= reps
print("s_memtime ticks per phase (rows = waves): prologue | stage | barrier | mfma loop | reduce   [n=%d chains=%d]" % (n, B))
for w in range(8):
    print("wave %d: " % w + "  ".join("%8.0f" % v for v in acc[w]) + "   total %8.0f" % acc[w].sum())
