"""PCIe-inclusive cost of the boundary: the same map2alm with host (numpy) and device (torch) buffers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, heracles_amd as hx
hx.init(0)
nside, lmax, nc = 4096, 6144, 4
plan = hx.Plan(nside, lmax)
rng = np.random.default_rng(1)
host = rng.standard_normal((nc, 12 * nside * nside))
dev = torch.as_tensor(host).cuda()
for name, m in (("device", dev), ("host", host)):
    plan.map2alm(m, 0)
    torch.cuda.synchronize(); t = time.perf_counter()
    a = plan.map2alm(m, 0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"map2alm {nc} spin-0 comps, {name} buffers: {dt*1e3:.1f} ms ({host.nbytes/1e9:.2f} GB of maps in, {nc*0.302:.2f} GB of alm out)")
pinned = torch.as_tensor(host).pin_memory()
plan.map2alm(pinned.numpy(), 0)
torch.cuda.synchronize(); t = time.perf_counter()
a = plan.map2alm(pinned.numpy(), 0)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"map2alm {nc} spin-0 comps, pinned host input: {dt*1e3:.1f} ms")
