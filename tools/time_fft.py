"""Ring Fourier stage time of one 8-component spin-0 map2alm (HX_LIBRARY selects the build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, heracles_amd as hx
hx.init(0)
nside, lmax = int(os.environ.get("NSIDE", 4096)), int(os.environ.get("LMAX", 6144))
plan = hx.Plan(nside, lmax)
m = torch.randn((8, 12 * nside * nside), dtype=torch.float64, device="cuda")
pw = torch.ones(12 * nside * nside, dtype=torch.float64, device="cuda") if os.environ.get("PW", "1") == "1" else None  # PW=0: without pixel weights
plan.map2alm(m, 0, pix_weights=pw)
hx._lib.profile_enable(True); hx._lib.profile_reset()
plan.map2alm(m, 0, pix_weights=pw)
print(os.environ.get("HX_LIBRARY", "default").split("/")[-1], "ring_fft ms (8 comps%s):" % (", pixel weights" if pw is not None else ""), round(hx._lib.profile_get("ring_fft")[1], 2))
