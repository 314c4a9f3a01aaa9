"""Where the time of one mixmat_eb (L = 6144) goes: host->host vs device output, per phase."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, heracles_amd as hx
hx.init(0)
L = int(os.environ.get("L", 6144))
ell = np.arange(L + 1)
wl = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / 3000.0) + 1e-3 / (1.0 + ell) ** 2
hx.mixmat_eb(wl[:65], l1max=64, l2max=64)
for rep in range(2):
    hx._lib.profile_enable(True); hx._lib.profile_reset()
    t = time.perf_counter(); mm = hx.mixmat_eb(wl); dt = time.perf_counter() - t
    print(f"mixmat_eb L={L} host->host: {dt*1e3:.1f} ms", {k: round(hx._lib.profile_get(k)[1], 2) for k in ("gauss_legendre", "wigner_tables", "weight_xi", "mixmat_gemm", "eb_combine")})
    hx._lib.profile_enable(False)
out = torch.empty((3, L + 1, L + 1), dtype=torch.float64, device="cuda")
L_ = hx._lib.load()
cl = np.ascontiguousarray(wl)
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    hx._lib.check(L_.hx_mixmat_eb(hx._lib.ptr(cl), len(cl), L, L, L, hx._lib.ptr(out)))
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"mixmat_eb L={L} device output: {dt*1e3:.1f} ms")
t = time.perf_counter(); h = out.cpu(); print(f"D2H of 3x{(L+1)**2*8/1e6:.0f} MB into fresh pageable memory: {(time.perf_counter()-t)*1e3:.1f} ms")
