"""Batched synthesis at the bench size: k_synth_duo (matrix unit) shapes against the vector-unit kernel (HX_SYNTH_KERNEL=valu in a
second process).  Prints the Legendre part, the table pass and the ring stage per sweep."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, heracles_amd as hx
hx.init(0)
nside, lmax = int(os.environ.get("NSIDE", 4096)), int(os.environ.get("LMAX", 6144))
plan = hx.Plan(nside, lmax)
nlm = (lmax + 1) * (lmax + 2) // 2
cases = [(2, 10), (0, 10), (2, 5), (2, 8), (0, 16), (0, 20)]
if os.environ.get("CASES"):
    cases = [tuple(int(v) for v in c.split(":")) for c in os.environ["CASES"].split(",")]
tag = os.environ.get("HX_SYNTH_KERNEL", "duo")
for spin, units in cases:
    n = units * (2 if spin else 1)
    alm = torch.randn((n, nlm), dtype=torch.complex128, device="cuda")
    out = torch.empty((n, 12 * nside * nside), dtype=torch.float64, device="cuda")
    plan.alm2map(alm, spin, out=out)
    res = []
    for rep in range(2):
        hx._lib.profile_enable(True); hx._lib.profile_reset()
        torch.cuda.synchronize(); t = time.perf_counter()
        plan.alm2map(alm, spin, out=out)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        hx._lib.profile_enable(False)
        res.append((dt * 1e3, {k: round(hx._lib.profile_get(k)[1], 1) for k in ("legendre_synthesis", "synth_table", "ring_fft")}))
    print(f"[{tag}] alm2map spin {spin} x {units}: " + " | ".join(f"{a:.1f} ms {b}" for a, b in res), flush=True)
    del alm, out
    torch.cuda.empty_cache()
if os.environ.get("NITER"):
    for spin, units in ((2, 10), (0, 10)):
        n = units * (2 if spin else 1)
        maps = torch.randn((n, 12 * nside * nside), dtype=torch.float64, device="cuda")
        a = torch.empty((n, nlm), dtype=torch.complex128, device="cuda")
        for rep in range(2):
            torch.cuda.synchronize(); t = time.perf_counter()
            plan.map2alm(maps, spin, niter=3, out=a)
            torch.cuda.synchronize(); print(f"[{tag}] map2alm niter=3 spin {spin} x {units} (call {rep}): {(time.perf_counter()-t)*1e3:.1f} ms", flush=True)
        del maps, a
        torch.cuda.empty_cache()
