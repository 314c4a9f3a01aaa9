"""Legendre synthesis time of one alm2map batch (SPIN, NCOMP; HX_LIBRARY selects the build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, heracles_amd as hx
hx.init(0)
nside, lmax = int(os.environ.get("NSIDE", 4096)), int(os.environ.get("LMAX", 6144))
spin, n = int(os.environ.get("SPIN", 2)), int(os.environ.get("NCOMP", 20))
plan = hx.Plan(nside, lmax)
nlm = (lmax + 1) * (lmax + 2) // 2
alm = torch.randn((n, nlm), dtype=torch.complex128, device="cuda")
out = torch.empty((n, 12 * nside * nside), dtype=torch.float64, device="cuda")
plan.alm2map(alm, spin, out=out)
hx._lib.profile_enable(True); hx._lib.profile_reset()
plan.alm2map(alm, spin, out=out)
torch.cuda.synchronize()
print(os.environ.get("HX_LIBRARY", "default").split("/")[-1], "spin", spin, "ncomp", n, "legendre_synthesis ms:", round(hx._lib.profile_get("legendre_synthesis")[1], 1))
