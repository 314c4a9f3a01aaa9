import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, heracles_amd as hx
hx.init(0)
nside, lmax = 8192, 8000
plan = hx.Plan(nside, lmax)
m = torch.randn((2, 12 * nside * nside), dtype=torch.float64, device="cuda")
plan.map2alm(m, 0)
hx._lib.profile_enable(True); hx._lib.profile_reset()
plan.map2alm(m, 0)
print("nside 8192, 2 spin-0 maps:", {k: round(hx._lib.profile_get(k)[1], 1) for k in ("ring_fft", "fourier_combine", "legendre_analysis", "alm_reduce")})
