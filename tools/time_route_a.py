"""Route A of INTEGRATION.md at the bench size with the reference's own data shapes: a dict of 20 numpy maps (10 bins x (POS spin 0,
SHE spin 2), nside 4096) -> heracles_amd.transform (one batched call, alms kept in HBM) -> heracles_amd.angular_power_spectra (210
map pairs) -> Result objects with numpy Cl blocks on the host."""
import os, sys, time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, heracles_amd as hx
hx.init(0)
nside, lmax = 4096, 6144
npix = 12 * nside * nside
mapper = hx.HipHealpixMapper(nside, lmax, deconvolve=False, niter=0)
mapper.pixel_weights = torch.ones(npix, dtype=torch.float64, device="cuda")
fields = {"POS": SimpleNamespace(spin=0, mapper_or_error=mapper), "SHE": SimpleNamespace(spin=2, mapper_or_error=mapper)}
g = torch.Generator(device="cuda").manual_seed(5)
maps = {}
for i in range(10):
    p = torch.randn((npix,), dtype=torch.float64, device="cuda", generator=g).cpu().numpy()
    s = torch.randn((2, npix), dtype=torch.float64, device="cuda", generator=g).cpu().numpy()
    hx.update_metadata(p, spin=0, fsky=1.0, musq=1.0, dens=1.0)
    hx.update_metadata(s, spin=2, fsky=1.0, musq=1.0, dens=1.0)
    maps["POS", i], maps["SHE", i] = p, s
for rep in range(3):
    t0 = time.perf_counter()
    alms = hx.transform(fields, maps, device="cuda")
    hx.synchronize()
    t1 = time.perf_counter()
    cls = hx.angular_power_spectra(alms)
    t2 = time.perf_counter()
    print(f"rep {rep}: transform {1e3 * (t1 - t0):.0f} ms + angular_power_spectra {1e3 * (t2 - t1):.0f} ms = {1e3 * (t2 - t0):.0f} ms; "
          f"{len(cls)} keys -> {210 / (t2 - t0):.0f} map pairs/s", flush=True)
