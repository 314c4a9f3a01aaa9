"""mixmat_eb (L = 6144) host -> host, eight builds in a row: the first fills fresh pages, the later ones recycled blocks of the host
result pool (heracles_amd/_lib.py); with HX_HOST_POOL_MB=0 every build fills fresh pages."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, heracles_amd as hx
hx.init(0)
L = 6144
ell = np.arange(L + 1)
wl = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / 3000.0) + 1e-3 / (1.0 + ell) ** 2
hx.mixmat_eb(wl[:65], l1max=64, l2max=64)
ts, addrs = [], []
mm = None
for rep in range(8):
    t = time.perf_counter(); mm = hx.mixmat_eb(wl); ts.append(time.perf_counter() - t)
    addrs.append(mm.__array_interface__["data"][0])
print("mixmat_eb host->host ms:", [round(x * 1e3, 1) for x in ts], "distinct blocks", len(set(addrs)), "checksum", float(np.abs(mm[2] - (mm[0] - mm[1])).max()))
