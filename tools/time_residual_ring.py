"""Ring-stage cost of a synthesis that writes the Jacobi residual (ref - synth) against a plain one, ten fields / ten maps at the bench size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, heracles_amd as hx
hx.init(0)
nside, lmax = 4096, 6144
plan = hx.Plan(nside, lmax)
nlm = (lmax + 1) * (lmax + 2) // 2
def fam(fn):
    fn(); torch.cuda.synchronize()
    hx._lib.profile_enable(True); hx._lib.profile_reset(); fn(); torch.cuda.synchronize()
    out = {k: round(hx._lib.profile_get(k)[1], 1) for k in ("ring_fft", "fourier_combine", "legendre_analysis", "legendre_synthesis")}
    hx._lib.profile_enable(False)
    return out
for spin, units in ((2, 10), (0, 10)):
    nc = units * (2 if spin else 1)
    m = torch.randn((nc, 12 * nside * nside), dtype=torch.float64, device="cuda")
    a = torch.empty((nc, nlm), dtype=torch.complex128, device="cuda")
    f0 = fam(lambda: plan.map2alm(m, spin, out=a, niter=0))
    f1 = fam(lambda: plan.map2alm(m, spin, out=a, niter=1))
    fs = fam(lambda: plan.alm2map(a, spin, out=m))
    print(f"spin {spin} x {units}: analysis ring {f0['ring_fft']}, niter=1 ring {f1['ring_fft']} -> residual synthesis ring {round(f1['ring_fft'] - 2 * f0['ring_fft'], 1)}; plain synthesis ring {fs['ring_fft']}", flush=True)
    del m, a; torch.cuda.empty_cache()
