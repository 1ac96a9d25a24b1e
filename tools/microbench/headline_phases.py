"""The budget of ONE graph-replayed headline launch (k_logpdf<4,1,2,2>: N = 256, 512 chains), from in-kernel s_memtime stamps kept
per launch (library: `make -C mcmc-date_amd/csrc stamp_headline` -> tools/microbench/libheadlinestamp.so; the sweep loop itself
carries no stamps).  The harness is bench.py's: graphs of 100 launches alternating two input batches, replayed back to back.

Per launch k (ring of the last 64): workgroup 0's milestones -- entry, prologue done (x requested and landed / chunk 0 in LDS),
first barrier, sweep done, ll stored -- and entry / exit of compute wave 0 of every workgroup 8 i (workgroup 0's XCD under the
round-robin placement).  Ticks are converted with the run's own rate: ticks between the entries of launches k and k + 32 over
32 x the HIP-event time per launch of the same replay."""
import ctypes as C
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
os.environ["MCD_LIB_PATH"] = os.path.join(HERE, os.environ.get("STAMPLIB", "libheadlinestamp.so"))
import mcmc_date_amd as M  # noqa: E402
from mcmc_date_amd import synthetic as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda", 0)
mu, sigma = S.random_spd_problem(n, seed=n)
lik = M.MvnLikelihood.from_covariance(mu, sigma, device=0)
X = [torch.as_tensor(S.sample_chains(mu, sigma, B, seed=n + 500 * i), device=dev) for i in range(2)]
ll = [torch.empty(B, dtype=torch.float64, device=dev) for _ in range(2)]
st = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    lik.logpdf_into(X[0], ll[0])
st.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st):
    for i in range(100):
        lik.logpdf_into(X[i & 1], ll[i & 1])
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 200
e0.record()
for _ in range(reps):
    g.replay()
e1.record()
torch.cuda.synchronize()
us_per_launch = e0.elapsed_time(e1) * 1e3 / (reps * 100)

L = M._capi.lib()
hist = np.zeros((64, 8, 8), np.uint64)
span = np.zeros((64, 64, 2), np.uint64)
nl = C.c_uint(0)
L.mcd_debug_hist.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint)]
assert L.mcd_debug_hist(hist.ctypes.data, span.ctypes.data, C.byref(nl)) == 0
hist = hist.astype(np.int64)
span = span.astype(np.int64)
last = nl.value                                     # launches completed; slot (last - 1) & 63 is the newest complete one
order = [(last - 64 + i) & 63 for i in range(64)]   # oldest ... newest
H = hist[order]
Sp = span[order]
nwg = min(64, (B // 2 + 7) // 8)                    # workgroups 8 i that exist (2 chains per workgroup)
entry = H[:, 0, 0]
period = np.diff(entry)
mid = slice(8, 56)                                  # inside one graph (a replay boundary may fall into the ring: dropped below)
per = period[mid]
ok = per < 2.0 * np.median(per)
ticks_per_launch = float(np.median(per[ok]))
ticks_per_us = ticks_per_launch / us_per_launch
print(f"N = {n}, {B} chains, graph replay: {us_per_launch:.3f} us per launch (HIP events over {reps * 100} launches); "
      f"{ticks_per_launch:.0f} s_memtime ticks per launch period -> {ticks_per_us:.1f} ticks per us")


def us(t):
    return t / ticks_per_us


rows = []
for k in range(8, 56):
    if not ok[k - 8]:
        continue
    e = H[k, 0, 0]
    # s_memtime is a per-XCD clock: a workgroup is compared with ITSELF only (workgroup 8 i of launch k and of launch k - 1 sit on one XCD
    # under the round-robin placement; across XCDs the counters differ by a constant of milliseconds)
    dur = Sp[k, :nwg, 1] - Sp[k, :nwg, 0]                           # entry -> exit of compute wave 0, per workgroup
    gap = Sp[k, :nwg, 0] - Sp[k - 1, :nwg, 1]                       # exit in launch k - 1 -> entry in launch k, per workgroup
    sane = (gap > 0) & (gap < 4 * ticks_per_launch) & (dur > 0) & (dur < 4 * ticks_per_launch)
    if sane.sum() < nwg // 2:
        continue
    rows.append(dict(
        gap=float(np.median(gap[sane])), gap_min=float(gap[sane].min()), gap_max=float(gap[sane].max()),
        dur_med=float(np.median(dur[sane])), dur_min=float(dur[sane].min()), dur_max=float(dur[sane].max()),
        c_prologue=H[k, 0, 1] - e, c_barrier=H[k, 0, 2] - e, c_sweep=H[k, 0, 3] - H[k, 0, 2], c_store=H[k, 0, 4] - H[k, 0, 3],
        l_entry=H[k, 2, 0] - e, l_chunk0=H[k, 2, 1] - H[k, 2, 0], l_barrier=H[k, 2, 2] - e, l_stream=H[k, 2, 3] - H[k, 2, 2],
        wg0=H[k, 0, 4] - e))
med = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
print(f"median over {len(rows)} launches, us (ticks):")
lines = [
    ("a workgroup's exit in the previous launch -> its entry in this one (kernel end, dispatch, launch of the waves): median over the workgroups 8 i", "gap"),
    ("   smallest / largest over those workgroups", "gap_min"), ("", "gap_max"),
    ("workgroup 0, compute wave 0: entry -> x, mu, 1/diag landed (one cold round trip)", "c_prologue"),
    ("   entry -> first barrier passed (chunk 0 of the factor in LDS)", "c_barrier"),
    ("   sweep of the 256 columns (32 chunks, one barrier each)", "c_sweep"),
    ("   sum of squares, DPP reduction, store", "c_store"),
    ("workgroup 0, loader wave 0: its entry after the compute wave's", "l_entry"),
    ("   chunk 0 requested -> written to LDS", "l_chunk0"),
    ("   stream of chunks 1 .. 31 (first barrier -> last barrier)", "l_stream"),
    ("workgroup 0 entry -> exit", "wg0"),
    ("entry -> exit of a workgroup: median / smallest / largest over the workgroups 8 i", "dur_med"), ("", "dur_min"), ("", "dur_max"),
]
for text, key in lines:
    print(f"  {us(med[key]):7.3f} us ({med[key]:7.0f})  {text}")
print(f"  gap + entry -> exit (medians) = {us(med['gap'] + med['dur_med']):.3f} us against the period {us_per_launch:.3f} us  (the stamps themselves cost about 1 us per launch: "
      "the same kernel without them runs the period in the bench line)")
