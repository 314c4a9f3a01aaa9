import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, heracles_amd as hx
hx.init(0)
nside, lmax = 4096, 6144
plan = hx.Plan(nside, lmax)
nlm = (lmax + 1) * (lmax + 2) // 2
fams = ("legendre_synthesis", "synth_spectrum", "synth_scatter", "ring_fft", "legendre_analysis", "fourier_combine", "alm_reduce")
for spin, n in ((0, 1), (2, 2), (0, 10), (2, 20)):
    alm = torch.randn((n, nlm), dtype=torch.complex128, device="cuda")
    out = torch.empty((n, 12 * nside * nside), dtype=torch.float64, device="cuda")
    plan.alm2map(alm, spin, out=out)
    hx._lib.profile_enable(True); hx._lib.profile_reset()
    torch.cuda.synchronize(); t = time.perf_counter()
    plan.alm2map(alm, spin, out=out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"alm2map spin {spin} {n} comps: {dt*1e3:.1f} ms", {k: round(hx._lib.profile_get(k)[1], 1) for k in fams}, flush=True)
    plan.map2alm(out, spin, niter=3)
    hx._lib.profile_reset()
    torch.cuda.synchronize(); t = time.perf_counter()
    plan.map2alm(out, spin, niter=3)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"map2alm niter=3 spin {spin} {n} comps: {dt*1e3:.1f} ms", {k: round(hx._lib.profile_get(k)[1], 1) for k in fams}, flush=True)
    del alm, out
