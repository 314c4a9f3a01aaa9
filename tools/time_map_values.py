"""Throughput of hx_map_values at nside=4096 (device-resident catalogue page and maps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, heracles_amd as hx
from heracles_amd.mapper import map_values, ang2pix_ring

hx.init(0)
nside = int(os.environ.get("NSIDE", 4096))
n = int(float(os.environ.get("NPOINTS", 1e8)))
npix = 12 * nside * nside
g = torch.Generator(device="cuda").manual_seed(1)
lon = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 360
lat = torch.rad2deg(torch.asin(torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1))
for nval in (1, 3):
    w = torch.randn((nval, n), dtype=torch.float64, device="cuda", generator=g)
    maps = torch.zeros((nval, npix), dtype=torch.float64, device="cuda")
    for ordered in (True, False):
        map_values(nside, lon, lat, maps, w, ordered=ordered)
        hx._lib.profile_enable(True); hx._lib.profile_reset()
        torch.cuda.synchronize(); t = time.perf_counter()
        map_values(nside, lon, lat, maps, w, ordered=ordered)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        parts = {k: round(hx._lib.profile_get(k)[1], 2) for k in ("ang2pix", "map_sort", "map_add")}
        hx._lib.profile_enable(False)
        print(f"map_values n={n:.0e} nval={nval} ordered={ordered}: {dt*1e3:.1f} ms = {n/dt/1e9:.2f} Gpoints/s", parts, flush=True)
ipix = torch.empty(n, dtype=torch.int64, device="cuda")
ang2pix_ring(nside, lon, lat, out=ipix)
torch.cuda.synchronize(); t = time.perf_counter()
ang2pix_ring(nside, lon, lat, out=ipix)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"ang2pix n={n:.0e}: {dt*1e3:.2f} ms = {24*n/dt/1e9:.0f} GB/s algorithmic (16 B in + 8 B out per point)")
