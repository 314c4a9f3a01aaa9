"""Where the time of a binned mixing-matrix request list goes (cProfile of heracles_amd.mixing_matrices over NKEYS keys)."""
import cProfile, os, pstats, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, heracles_amd as hx
hx.init(0)
L = int(os.environ.get("L", 6144)); nb = int(os.environ.get("NBINS", 4))
ell = np.arange(L + 1)
fields = {"POS": types.SimpleNamespace(mask="VIS", spin=0), "SHE": types.SimpleNamespace(mask="WHT", spin=2), "CON": types.SimpleNamespace(mask="WHT", spin=0)}
mcls = {}
for a, b in (("VIS", "VIS"), ("VIS", "WHT"), ("WHT", "WHT")):
    for i in range(nb):
        for j in range(i if a == b else 0, nb):
            mcls[a, b, i, j] = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / (3000.0 + 40.0 * i + 7.0 * j)) + 1e-3 / (1.0 + ell) ** 2
edges = np.unique(np.geomspace(2, L + 1, 33).astype(int))
kw = dict(l1max=L, l2max=L, l3max=L, bins=edges, weights="2l+1")
hx.mixing_matrices(fields, {k: mcls[k] for k in list(mcls)[:2]}, **kw)
t = time.perf_counter(); out = hx.mixing_matrices(fields, mcls, **kw); dt = time.perf_counter() - t
print(f"{len(out)} keys in {dt:.3f} s = {dt / len(out) * 1e3:.2f} ms per key")
pr = cProfile.Profile(); pr.enable(); hx.mixing_matrices(fields, mcls, **kw); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
os.environ["HX_MIXMAT_TRACE"] = "1"
