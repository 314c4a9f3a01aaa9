"""Diagnostic driver: a few spin-0 / spin-2 analysis launches at nside 2048 (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import heracles_amd as hx
hx.init(0)
nside, lmax = int(os.environ.get("NSIDE", 2048)), int(os.environ.get("LMAX", 3072))
spin = int(os.environ.get("SPIN", 0))
plan = hx.Plan(nside, lmax)
m = torch.randn((int(os.environ.get("NCOMP", 8)), 12 * nside * nside), dtype=torch.float64, device="cuda")
for _ in range(2):
    plan.map2alm(m, spin)
hx._lib.profile_enable(True); hx._lib.profile_reset()
plan.map2alm(m, spin)
n, ms = hx._lib.profile_get("legendre_analysis")
print("spin", spin, "legendre ms/launch", ms / n, "launches", n, "total ms", ms)
