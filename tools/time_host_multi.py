"""Host -> host leg of the bench as bench.py runs it (one hx_map2alm_multi call for 10 spin-2 + 10 spin-0 maps from pageable
numpy arrays), with the library's timeline (HX_TRACE=1) and the kernel families' HIP-event times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HX_TRACE", "1")
import numpy as np, torch, heracles_amd as hx
hx.init(0)
nside, lmax = 4096, 6144
npix, nlm = 12 * nside * nside, (lmax + 1) * (lmax + 2) // 2
plan = hx.Plan(nside, lmax)
h2 = torch.randn((20, npix), dtype=torch.float64, device="cuda").cpu().numpy()
h0 = torch.randn((10, npix), dtype=torch.float64, device="cuda").cpu().numpy()
o2 = torch.empty((20, nlm), dtype=torch.complex128, device="cuda")
o0 = torch.empty((10, nlm), dtype=torch.complex128, device="cuda")
pw = torch.ones(npix, dtype=torch.float64, device="cuda")
for rep in range(3):
    hx._lib.profile_enable(True); hx._lib.profile_reset()
    torch.cuda.synchronize(); t = time.perf_counter()
    plan.map2alm_multi([(h2, 2, o2), (h0, 0, o0)], pix_weights=pw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    fam = {k: round(hx._lib.profile_get(k)[1], 1) for k in ("ring_fft", "fourier_combine", "legendre_analysis", "alm_reduce")}
    print(f"rep {rep}: {dt*1e3:.0f} ms; kernels {fam}", flush=True)
