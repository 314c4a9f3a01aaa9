"""Throughput of the point transform (hx_pointsht_adjoint) at the bench's band limit: device-resident points."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, heracles_amd as hx

hx.init(0)
lmax = int(os.environ.get("LMAX", 6144))
sht = hx.PointSHT(lmax)
print("lmax", lmax, "rings on the circle", sht.nrings_circle, "grid", sht.ngrid, "kernel width", sht.kernel_width, flush=True)
g = torch.Generator(device="cuda").manual_seed(1)
for n in [int(float(x)) for x in os.environ.get("NPOINTS", "1e6,1e7,1e8").split(",")]:
    loc = torch.empty((n, 2), dtype=torch.float64, device="cuda")
    loc[:, 0] = torch.acos(torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1)
    loc[:, 1] = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 6.283185307179586
    for spin, nc in ((0, 1), (2, 2)):
        v = torch.randn((nc, n), dtype=torch.float64, device="cuda", generator=g)
        out = sht.adjoint_synthesis(loc, v, spin=spin)
        hx._lib.profile_enable(True); hx._lib.profile_reset()
        torch.cuda.synchronize(); t = time.perf_counter()
        sht.adjoint_synthesis(loc, v, spin=spin, out=out)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        parts = {k: round(hx._lib.profile_get(k)[1], 1) for k in ("nufft_spread", "nufft_fft", "fourier_combine", "legendre_analysis", "alm_reduce")}
        hx._lib.profile_enable(False)
        print(f"n={n:.0e} spin {spin}: {dt*1e3:.1f} ms = {n/dt/1e6:.1f} Mpoints/s", parts, flush=True)
