"""Batched synthesis at nside 8192 / lmax 8000: a sweep of UNITS fields (default 6) needs 128 + 77 GB of ring modes and spectra at full
width -- the library cuts it to the HBM that is free.  First and last field against the single-field (vector-unit) sweeps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, heracles_amd as hx
hx.init(0)
nside, lmax = 8192, 8000
units = int(os.environ.get("UNITS", 6))
plan = hx.Plan(nside, lmax)
nlm = (lmax + 1) * (lmax + 2) // 2
g = torch.Generator(device="cuda").manual_seed(3)
alm = torch.randn((2 * units, nlm), dtype=torch.complex128, device="cuda", generator=g)
alm[:, : lmax + 1] = alm[:, : lmax + 1].real.to(torch.complex128)
idx = torch.arange(nlm, device="cuda")
for m in range(2):  # l < 2 has no spin-2 content
    for l in range(m, 2):
        alm[:, m * (2 * lmax + 1 - m) // 2 + l] = 0
out = torch.empty((2 * units, 12 * nside * nside), dtype=torch.float64, device="cuda")
free0 = torch.cuda.mem_get_info()[0] / 1e9
t = time.perf_counter(); plan.alm2map(alm, 2, out=out); torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"alm2map of {units} fields at nside {nside}: {dt:.2f} s (free HBM before the call {free0:.0f} GB, after {torch.cuda.mem_get_info()[0] / 1e9:.0f} GB)", flush=True)
one = torch.empty((2, 12 * nside * nside), dtype=torch.float64, device="cuda")
scale = max(float(out[i].abs().max()) for i in range(out.shape[0]))  # (row by row: the library holds most of the HBM as scratch)
for u in (0, units - 1):
    plan.alm2map(alm[2 * u : 2 * u + 2], 2, out=one)
    one -= out[2 * u : 2 * u + 2]
    print(f"field {u}: max |batch - single| / max = {float(one.abs_().max()) / scale:.2e}", flush=True)
