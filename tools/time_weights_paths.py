import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, heracles_amd as hx
from heracles_amd import weights as hxw
hx.init(0)
nside, lmax = 4096, 6144
plan = hx.Plan(nside, lmax)
m = torch.randn((10, 12 * nside * nside), dtype=torch.float64, device="cuda")
out = torch.empty((10, (lmax + 1) * (lmax + 2) // 2), dtype=torch.complex128, device="cuda")
pws = {"none": None, "generic": 1.0 + 0.01 * torch.rand(12 * nside * nside, dtype=torch.float64, device="cuda"),
       "symmetric": hxw.expand_pixel_weights(nside, 1e-3 * np.random.default_rng(7).standard_normal(hxw.compressed_size(nside)), device="cuda")}
for name, pw in pws.items():
    plan.map2alm(m, 0, pix_weights=pw, out=out)
    torch.cuda.synchronize()
    hx._lib.profile_enable(True); hx._lib.profile_reset()
    t = time.perf_counter()
    for _ in range(3):
        plan.map2alm(m, 0, pix_weights=pw, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3 * 1e3
    fam = {k: round(hx._lib.profile_get(k)[1] / 3, 2) for k in ("ring_fft", "fourier_combine", "legendre_analysis", "alm_reduce")}
    print(name, f"wall {dt:.2f} ms", fam, "sum", round(sum(fam.values()), 2), flush=True)
