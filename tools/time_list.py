"""The reference's data layout -- one numpy array per map -- through hx_map2alm_list (no stacked copy) against np.stack + hx_map2alm_multi:
10 spin-2 fields + 10 spin-0 maps at nside 4096, pixel weights applied; alms to device arrays and to numpy arrays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, heracles_amd as hx
hx.init(0)
nside, lmax = 4096, 6144
npix, nlm = 12 * nside * nside, (lmax + 1) * (lmax + 2) // 2
plan = hx.Plan(nside, lmax)
g = torch.Generator(device="cuda").manual_seed(3)
maps = [torch.randn((2, npix), dtype=torch.float64, device="cuda", generator=g).cpu().numpy() for _ in range(10)]
maps += [torch.randn((npix,), dtype=torch.float64, device="cuda", generator=g).cpu().numpy() for _ in range(10)]
spins = [2] * 10 + [0] * 10
pw = torch.ones(npix, dtype=torch.float64, device="cuda")
dev_out = [torch.empty((2, nlm) if s else (nlm,), dtype=torch.complex128, device="cuda") for s in spins]
for rep in range(2):
    t = time.perf_counter()
    plan.map2alm_list(maps, spins, outs=dev_out, pix_weights=pw)
    torch.cuda.synchronize()
    print(f"list -> device alms: {(time.perf_counter() - t) * 1e3:.0f} ms", flush=True)
for rep in range(2):
    t = time.perf_counter()
    out = plan.map2alm_list(maps, spins, pix_weights=pw)
    print(f"list -> numpy alms: {(time.perf_counter() - t) * 1e3:.0f} ms", flush=True)
t = time.perf_counter()
s2, s0 = np.stack(maps[:10]).reshape(20, npix), np.stack(maps[10:])
t1 = time.perf_counter()
o2 = torch.empty((20, nlm), dtype=torch.complex128, device="cuda"); o0 = torch.empty((10, nlm), dtype=torch.complex128, device="cuda")
plan.map2alm_multi([(s2, 2, o2), (s0, 0, o0)], pix_weights=pw)
torch.cuda.synchronize()
print(f"np.stack {(t1 - t) * 1e3:.0f} ms + multi {(time.perf_counter() - t1) * 1e3:.0f} ms", flush=True)
a = np.asarray(out[3]); b = o2[6:8].cpu().numpy()
print("agree:", float(np.abs(a - b).max() / np.abs(b).max()))
