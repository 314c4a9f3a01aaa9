"""alm2cl of the bench's 30 components (475 spectra) at lmax 6144; HX_LIBRARY selects the build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, heracles_amd as hx
from heracles_amd import distributed as hxd
hx.init(0)
lmax, nb = 6144, 10
nlm = (lmax + 1) * (lmax + 2) // 2
g = torch.Generator(device="cuda").manual_seed(1)
a0 = torch.view_as_complex(torch.randn((nb, nlm, 2), dtype=torch.float64, device="cuda", generator=g))
a2 = torch.view_as_complex(torch.randn((nb, 2, nlm, 2), dtype=torch.float64, device="cuda", generator=g))
work = hxd.PairWork(1, 0, nb, nlm, lmax)
cls = work.all_pairs_cl(a0, a2)
hx._lib.profile_enable(True); hx._lib.profile_reset()
torch.cuda.synchronize(); t = time.perf_counter()
cls = work.all_pairs_cl(a0, a2)
torch.cuda.synchronize(); dt = time.perf_counter() - t
chk = float(sum(np.asarray(v).sum() for v in cls.values())) if hasattr(cls, "values") else 0.0
print(os.environ.get("HX_LIBRARY", "default").split("/")[-1], f"all_pairs_cl {dt*1e3:.1f} ms, kernel {hx._lib.profile_get('alm2cl')[1]:.2f} ms, checksum {chk:.6e}")
