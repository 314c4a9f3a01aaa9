"""nside 8192 / lmax 8000 (examples/heracles.cfg:56-62) through the batched kernels: five spin-0 maps and three spin-2 fields per call, per-family kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, heracles_amd as hx
hx.init(0)
nside, lmax = 8192, 8000
plan = hx.Plan(nside, lmax)
nlm = (lmax + 1) * (lmax + 2) // 2
for spin, units in ((0, 5), (2, 3)):
    nc = units * (2 if spin else 1)
    m = torch.randn((nc, 12 * nside * nside), dtype=torch.float64, device="cuda")
    a = torch.empty((nc, nlm), dtype=torch.complex128, device="cuda")
    for what, fn in (("map2alm niter=0", lambda: plan.map2alm(m, spin, out=a)), ("alm2map", lambda: plan.alm2map(a, spin, out=m))):
        fn(); torch.cuda.synchronize()
        hx._lib.profile_enable(True); hx._lib.profile_reset()
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t
        fam = {k: round(hx._lib.profile_get(k)[1], 1) for k in ("ring_fft", "fourier_combine", "legendre_analysis", "legendre_synthesis", "synth_table", "alm_reduce") if hx._lib.profile_get(k)[0]}
        hx._lib.profile_enable(False)
        print(f"nside {nside} lmax {lmax} spin {spin} x {units}: {what} {dt * 1e3:.0f} ms {fam}", flush=True)
    del m, a
    torch.cuda.empty_cache()
