"""Host -> host leg of the bench in pieces: map2alm of the bench's batches with pageable numpy maps vs device tensors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, heracles_amd as hx
hx.init(0)
nside, lmax = 4096, 6144
npix, nlm = 12 * nside * nside, (lmax + 1) * (lmax + 2) // 2
plan = hx.Plan(nside, lmax)
for spin, nc in ((0, 10), (2, 20)):
    dev = torch.randn((nc, npix), dtype=torch.float64, device="cuda")
    host = dev.cpu().numpy()
    out = torch.empty((nc, nlm), dtype=torch.complex128, device="cuda")
    for name, m in (("device", dev), ("host", host)):
        plan.map2alm(m, spin, out=out)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            plan.map2alm(m, spin, out=out)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        print(f"spin {spin}, {nc} components ({host.nbytes/1e9:.1f} GB), {name} maps: {min(ts)*1e3:.0f} ms (runs: {' '.join('%.0f' % (x*1e3) for x in ts)})", flush=True)
    del dev, host, out
    torch.cuda.empty_cache()
