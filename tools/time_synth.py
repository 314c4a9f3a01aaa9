import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, heracles_amd as hx
hx.init(0)
nside, lmax = int(os.environ.get("NSIDE", 4096)), int(os.environ.get("LMAX", 6144))
plan = hx.Plan(nside, lmax)
nlm = (lmax + 1) * (lmax + 2) // 2
for spin, n in ((0, 8), (2, 8)):
    alm = torch.randn((n, nlm), dtype=torch.complex128, device="cuda")
    out = torch.empty((n, 12 * nside * nside), dtype=torch.float64, device="cuda")
    plan.alm2map(alm, spin, out=out)
    hx._lib.profile_enable(True); hx._lib.profile_reset()
    torch.cuda.synchronize(); t = time.perf_counter()
    plan.alm2map(alm, spin, out=out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"alm2map spin {spin} {n} comps: {dt*1e3:.1f} ms", {k: round(hx._lib.profile_get(k)[1], 1) for k in ("legendre_synthesis", "ring_fft")})
    for rep in range(2):  # the first call grows the plan's scratch buffers
        t = time.perf_counter()
        a2 = plan.map2alm(out, spin, niter=1)
        torch.cuda.synchronize(); print(f"map2alm niter=1 (call {rep}): {(time.perf_counter()-t)*1e3:.1f} ms")
