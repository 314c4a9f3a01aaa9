"""Leak / stability check of the binned mixing-matrix path: contexts created and destroyed in a loop, bins reset, results into every kind of destination."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, heracles_amd as hx
from heracles_amd.binning import BinPlan
hx.init(0)
L = int(os.environ.get("L", 2048))
ell = np.arange(L + 1)
wl = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / 3000.0) + 1e-3 / (1.0 + ell) ** 2
ref = None
free0 = None
for rep in range(int(os.environ.get("REPS", 60))):
    nb = 8 + (rep % 5) * 13
    edges = np.unique(np.geomspace(2, L + 1, nb + 1).astype(int))
    with hx.MixmatContext(L, L - 100, L + 50) as ctx:
        for weights in (None, "2l+1"):
            ctx.set_bins(BinPlan(ell, edges, weights))
            for spin in ((0, 0), (0, 2), (2, 2)):
                a = ctx.binned(wl, spin)
                b = ctx.binned(wl, spin, out=torch.empty(a.shape, dtype=torch.float64, device="cuda")).cpu().numpy()
                c = ctx.binned(wl, spin, out=hx.pinned_empty(a.shape))
                assert np.array_equal(a, b) and np.array_equal(a, c)
                if rep % 5 == 0 and weights == "2l+1" and spin == (2, 2):
                    if ref is None: ref = a.copy()
                    assert np.array_equal(ref, a), "not repeatable across contexts"
    torch.cuda.synchronize()
    if rep == 4: free0 = torch.cuda.mem_get_info()[0]
    if rep % 10 == 9: print(f"rep {rep}: free HBM {torch.cuda.mem_get_info()[0] / 1e9:.2f} GB", flush=True)
hx.release_caches()
free1 = torch.cuda.mem_get_info()[0]
print("free HBM after rep 4:", free0 / 1e9, "GB; at the end:", free1 / 1e9, "GB")
assert free1 >= free0 - 64e6, "HBM leak"
print("soak ok")
