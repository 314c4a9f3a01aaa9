"""Matrix / vector flops the Legendre kernels execute in one hx_map2alm (the kernels' own counters); HX_LIBRARY selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, heracles_amd as hx
hx.init(0)
nside, lmax = int(os.environ.get("NSIDE", 4096)), int(os.environ.get("LMAX", 6144))
spin, nc = int(os.environ.get("SPIN", 2)), int(os.environ.get("NCOMP", 20))
plan = hx.Plan(nside, lmax)
m = torch.randn((nc, 12 * nside * nside), dtype=torch.float64, device="cuda")
plan.map2alm(m, spin)
hx._lib.executed_flops(reset=True)
plan.map2alm(m, spin)
torch.cuda.synchronize()
print(os.environ.get("HX_LIBRARY", "default").split("/")[-1], "spin", spin, "ncomp", nc, "executed (matrix, vector) flops:", hx._lib.executed_flops(reset=True))
