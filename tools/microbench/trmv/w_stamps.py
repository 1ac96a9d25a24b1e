import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MCD_LIB_PATH"] = os.path.join(ROOT, "tools", "microbench", "libwstamp.so")
os.environ["MCD_W"] = "1"
sys.path.insert(0, ROOT)
import numpy as np, torch
import mcmc_date_amd as M
from mcmc_date_amd import synthetic as S
n, B = int(sys.argv[1]), int(sys.argv[2])
mu, sigma = S.random_spd_problem(n, seed=1)
X = torch.as_tensor(S.sample_chains(mu, sigma, B, seed=1), device="cuda")
lik = M.MvnLikelihood.from_covariance(mu, sigma)
for _ in range(20): ll = lik.logpdf(X)
torch.cuda.synchronize()
out = (C.c_ulonglong * 128)()
M._capi.lib().mcd_w_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
M._capi.lib().mcd_w_debug_stamps(out)
a = np.array(out[:]).reshape(16, 8)[:, :5]
print("per wave cycles [prefetch+stage, barrier, pass, barrier, reduce]:")
print(a[:int(os.environ.get("MCD_W_NW", "8"))])
