import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, heracles_amd as hx
hx.init(0)
L = 6144; nb = 13
ell = np.arange(L + 1)
fields = {"POS": types.SimpleNamespace(mask="VIS", spin=0), "SHE": types.SimpleNamespace(mask="WHT", spin=2), "CON": types.SimpleNamespace(mask="WHT", spin=0)}
mcls = {}
for a, b in (("VIS", "VIS"), ("VIS", "WHT"), ("WHT", "WHT")):
    for i in range(nb):
        for j in range(i if a == b else 0, nb):
            mcls[a, b, i, j] = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / (3000.0 + 40.0 * i + 7.0 * j)) + 1e-3 / (1.0 + ell) ** 2
edges = np.unique(np.geomspace(2, L + 1, 33).astype(int))
kw = dict(l1max=L, l2max=L, l3max=L, bins=edges, weights="2l+1")
def run(tag):
    hx.mixing_matrices(fields, {k: mcls[k] for k in list(mcls)[:2]}, **kw)
    hx._lib.profile_enable(True); hx._lib.profile_reset()
    t = time.perf_counter(); out = hx.mixing_matrices(fields, mcls, **kw); dt = time.perf_counter() - t
    fam = {k: round(hx._lib.profile_get(k)[1], 1) for k in ("wigner_tables", "mixmat_bin_table", "weight_xi", "mixmat_binned")}
    hx._lib.profile_enable(False)
    print(f"{tag}: {len(out)} keys in {dt:.3f} s = {dt / len(out) * 1e3:.2f} ms per key; kernel families, ms in all: {fam}; free HBM {torch.cuda.mem_get_info()[0] / 1e9:.0f} GB", flush=True)
run("fresh process")
big = torch.empty(int(150e9 // 8), dtype=torch.float64, device="cuda"); big.zero_(); torch.cuda.synchronize()
run("150 GB of HBM held by torch")
del big; torch.cuda.empty_cache()
plan = hx.Plan(4096, 6144)
m = torch.randn((4, 12 * 4096 * 4096), dtype=torch.float64, device="cuda")
a = plan.map2alm(m, 0)
run("after a map2alm (plan scratch held)")
h = m.cpu().numpy()
a2 = plan.map2alm(h, 0)
run("after a host-map map2alm (pinned staging buffers held)")
plan.release_scratch()
run("after release_scratch")
