"""mixmat_eb (L = 6144), eight builds in a row per destination: a fresh numpy array (the default), a pageable and a page-locked `out=` the
caller re-uses, and MixmatContext with its own result buffer (the path of heracles_amd.mixing_matrices)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, heracles_amd as hx
hx.init(0)
L = int(os.environ.get("L", 6144))
ell = np.arange(L + 1)
wl = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / 3000.0) + 1e-3 / (1.0 + ell) ** 2
hx.mixmat_eb(wl[:65], l1max=64, l2max=64)
shape = (3, L + 1, L + 1)


def row(name, fn):
    ts = []
    for rep in range(8):
        t = time.perf_counter(); mm = fn(); ts.append(time.perf_counter() - t)
    print(f"{name}: ms", [round(x * 1e3, 1) for x in ts], "checksum", float(np.abs(mm[2] - (mm[0] - mm[1])).max()), flush=True)


row("fresh numpy (default)", lambda: hx.mixmat_eb(wl))
page = np.zeros(shape)
row("out= pageable, re-used", lambda: hx.mixmat_eb(wl, out=page))
pin = hx.pinned_empty(shape)
row("out= page-locked, re-used", lambda: hx.mixmat_eb(wl, out=pin))
with hx.MixmatContext(L, L, L) as ctx:
    rb = ctx.result_buffer((2, 2))
    row("MixmatContext + result_buffer", lambda: ctx(wl, (2, 2), out=rb))
    row("MixmatContext, fresh numpy", lambda: ctx(wl, (2, 2)))
