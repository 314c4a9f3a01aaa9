"""Context created AFTER a large HBM allocation: kernel families of a binned key."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, heracles_amd as hx
from heracles_amd.binning import BinPlan
hx.init(0)
L = 6144
ell = np.arange(L + 1)
wl = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / 3000.0) + 1e-3 / (1.0 + ell) ** 2
edges = np.unique(np.geomspace(2, L + 1, 33).astype(int))
def report(tag):
    ctx = hx.MixmatContext(L, L, L); ctx.set_bins(BinPlan(ell, edges, "2l+1"))
    for spin in ((0, 0), (2, 2)):
        ctx.binned(wl, spin)
        hx._lib.profile_enable(True); hx._lib.profile_reset()
        t0 = time.perf_counter()
        for _ in range(20): ctx.binned(wl, spin)
        dt = (time.perf_counter() - t0) / 20 * 1e3
        fam = {k: round(hx._lib.profile_get(k)[1] / 20, 3) for k in ("weight_xi", "mixmat_binned")}
        hx._lib.profile_enable(False)
        print(f"{tag} spin {spin}: {dt:.2f} ms per key, kernel ms {fam}", flush=True)
    ctx.close()
report("fresh process")
big = torch.empty(int(150e9 // 8), dtype=torch.float64, device="cuda"); big.zero_(); torch.cuda.synchronize()
report("context created with 150 GB held")
del big; torch.cuda.empty_cache(); torch.cuda.synchronize()
report("context created after the 150 GB were freed")
