"""Diagnostic: milestone timestamps of workgroup 0 (needs tools/microbench/libstamp.so built with -DMCD_STAMP)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mcmc_date_amd import synthetic as S
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("STAMPLIB", "libstamp.so")))
dp = C.POINTER(C.c_double)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
mu, sigma = S.random_spd_problem(n, seed=n)
X = torch.as_tensor(S.sample_chains(mu, sigma, B, seed=1), device="cuda")
ll = torch.empty(B, dtype=torch.float64, device="cuda")
h = C.c_void_p()
L.mcd_mvn_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, dp, dp, C.c_int, C.c_double, C.c_int]
assert L.mcd_mvn_create(C.byref(h), n, mu.ctypes.data_as(dp), np.ascontiguousarray(sigma).ctypes.data_as(dp), 0, 0.0, 0) == 0
L.mcd_mvn_logpdf_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
for _ in range(20):
    L.mcd_mvn_logpdf_batch(h, X.data_ptr(), n, B, 1, None, ll.data_ptr())
torch.cuda.synchronize()
out = (C.c_ulonglong * 64)()
L.mcd_debug_stamps(out)
t = np.array(list(out), dtype=np.int64).reshape(8, 8)
t0 = t[:, 0][t[:, 0] > 0].min()
names = ["entry", "prologue done", "first barrier", "sweep done", "ll stored"]
for w in range(8):
    if t[w, 0] == 0:
        continue
    print(f"wave {w}: " + "  ".join(f"{names[i]}={t[w, i] - t0}" for i in range(5) if t[w, i] > 0))
print("accumulated in the sweep (cycles): compute waves: [apply, barrier wait, -]; loaders: [store(+vmcnt wait), barrier wait, load issue]")
for w in range(8):
    if t[w, 0]:
        print(f"wave {w}:", t[w, 5:8])
