"""Which part of a binned mixing-matrix key slows down (0.6 -> 3.7 ms) once the process has held a large HBM allocation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, heracles_amd as hx
from heracles_amd.binning import BinPlan
hx.init(0)
L = 6144
ell = np.arange(L + 1)
wl = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / 3000.0) + 1e-3 / (1.0 + ell) ** 2
edges = np.unique(np.geomspace(2, L + 1, 33).astype(int))
ctx = hx.MixmatContext(L, L, L); ctx.set_bins(BinPlan(ell, edges, "2l+1"))
pin = hx.pinned_empty((31, L + 1)); dev = torch.empty((31, L + 1), dtype=torch.float64, device="cuda"); page = np.zeros((31, L + 1))
wl_dev = torch.as_tensor(wl).cuda()
def t(fn, n=50):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def report(tag):
    print(f"{tag}: fresh numpy {t(lambda: ctx.binned(wl, (0, 0))):.2f} ms | pageable out= {t(lambda: ctx.binned(wl, (0, 0), out=page)):.2f} | pinned out= {t(lambda: ctx.binned(wl, (0, 0), out=pin)):.2f} | "
          f"device out= {t(lambda: ctx.binned(wl, (0, 0), out=dev)):.2f} | device cl + device out {t(lambda: ctx.binned_dev(wl_dev, (0, 0), dev) if hasattr(ctx, 'binned_dev') else ctx.binned(wl, (0, 0), out=dev)):.2f} | "
          f"torch D2H 1.5 MB pageable {t(lambda: dev.cpu()):.2f} | torch H2D 49 KB {t(lambda: torch.as_tensor(wl).cuda()):.2f} | np.empty+touch {t(lambda: np.empty((31, L + 1)).fill(0)):.2f}", flush=True)
report("fresh process")
big = torch.empty(int(150e9 // 8), dtype=torch.float64, device="cuda"); big.zero_(); torch.cuda.synchronize()
report("150 GB held")
del big; torch.cuda.empty_cache(); torch.cuda.synchronize()
report("150 GB freed")
