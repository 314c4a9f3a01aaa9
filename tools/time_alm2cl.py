"""alm2cl of the bench's 30 components (475 spectra) at lmax 6144; HX_LIBRARY selects the build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, heracles_amd as hx
hx.init(0)
lmax = int(os.environ.get("LMAX", 6144)); ncomp = int(os.environ.get("NCOMP", 30))
nlm = (lmax + 1) * (lmax + 2) // 2
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.view_as_complex(torch.randn((ncomp, nlm, 2), dtype=torch.float64, device="cuda", generator=g))
comps = [a[k] for k in range(ncomp)]
pairs = [(i, j) for i in range(ncomp) for j in range(i, ncomp)]
cls = hx.alm2cl_pairs(comps, pairs, lmax)
hx._lib.profile_enable(True); hx._lib.profile_reset()
torch.cuda.synchronize(); t = time.perf_counter()
cls = hx.alm2cl_pairs(comps, pairs, lmax)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print(os.environ.get("HX_LIBRARY", "default").split("/")[-1], f"{ncomp} components, {len(pairs)} spectra: call {dt*1e3:.1f} ms, kernel {hx._lib.profile_get('alm2cl')[1]:.2f} ms, "
      f"minimal traffic {ncomp * nlm * 16 / 1e9:.1f} GB, checksum {float(np.abs(cls).sum()):.6e}")
