"""One binned mixing-matrix key (the production call: bins = 32 log 2l+1) at L = 6144, host -> host, per kind; kernel families."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, heracles_amd as hx
from heracles_amd.binning import BinPlan
hx.init(0)
L = int(os.environ.get("L", 6144))
NB = int(os.environ.get("NB", 32))
ell = np.arange(L + 1)
wl = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / 3000.0) + 1e-3 / (1.0 + ell) ** 2
edges = np.unique(np.geomspace(2, L + 1, NB + 1).astype(int))
t = time.perf_counter()
ctx = hx.MixmatContext(L, L, L)
plan = BinPlan(ell, edges, "2l+1")
ctx.set_bins(plan)
print(f"context + set_bins: {(time.perf_counter()-t)*1e3:.1f} ms, {plan.nbins} bins")
for spin in ((0, 0), (0, 2), (2, 2)):
    out = None
    for rep in range(5):
        hx._lib.profile_enable(True); hx._lib.profile_reset()
        t = time.perf_counter(); out = ctx.binned(wl, spin, out=out); dt = time.perf_counter() - t
        fam = {k: round(hx._lib.profile_get(k)[1], 3) for k in ("wigner_tables", "mixmat_bin_table", "mixmat_binned")}
        hx._lib.profile_enable(False)
        print(f"binned spin {spin} L={L}: {dt*1e3:.2f} ms host->host", fam)
ctx.close()
