"""Diagnostic driver: one timed hx_map2alm (after two warm-up calls) with the per-family kernel times of the library's own
HIP events.  NSIDE / LMAX / SPIN / NCOMP from the environment (defaults: nside 2048, lmax 3072, spin 0, 8 components);
HX_LIBRARY selects another build of the library (A/B on one device)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import heracles_amd as hx
hx.init(0)
nside, lmax = int(os.environ.get("NSIDE", 2048)), int(os.environ.get("LMAX", 3072))
spin = int(os.environ.get("SPIN", 0))
plan = hx.Plan(nside, lmax)
m = torch.randn((int(os.environ.get("NCOMP", 8)), 12 * nside * nside), dtype=torch.float64, device="cuda")
pw = torch.ones(12 * nside * nside, dtype=torch.float64, device="cuda") if os.environ.get("PW") == "1" else None  # PW=1: with pixel weights
for _ in range(2):
    plan.map2alm(m, spin, pix_weights=pw)
hx._lib.profile_enable(True); hx._lib.profile_reset()
plan.map2alm(m, spin, pix_weights=pw)
out = []
for fam in ("ring_fft", "fourier_combine", "legendre_analysis", "alm_reduce"):
    n, ms = hx._lib.profile_get(fam)
    out.append("%s %.2f ms / %d" % (fam, ms, n))
print("spin", spin, "ncomp", m.shape[0], "pw" if pw is not None else "", "|", " | ".join(out), "| lib", os.environ.get("HX_LIBRARY", "default"))
