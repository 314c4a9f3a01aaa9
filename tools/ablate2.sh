#!/bin/bash
# spin-0 only diagnostic (nbins comps in one launch): recursion-phase cost breakdown
for a in 0 1 17 33 49 5 21 53; do
  HX_ABLATE=$a python - <<PY
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, heracles_amd as hx
hx.init(0)
nside, lmax = 2048, 3072
plan = hx.Plan(nside, lmax)
m = torch.randn((8, 12*nside*nside), dtype=torch.float64, device="cuda")
plan.map2alm(m, 0)
hx._lib.profile_enable(True); hx._lib.profile_reset()
plan.map2alm(m, 0); plan.map2alm(m, 0)
n, ms = hx._lib.profile_get("legendre_analysis")
print("ablate $a spin0 legendre ms/launch", round(ms/n, 2))
PY
done
