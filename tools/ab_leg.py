"""A/B of Legendre-analysis builds on one device at the bench size: times of the kernel families for ten spin-2 fields and ten spin-0
maps, and a strided sample of the alms written to gpurun_out/ablag_<TAG>_<spin>.npy (compare two tags with COMPARE=tagA,tagB).
HX_LIBRARY selects the build (tools/build_variant.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("COMPARE"):
    a, b = os.environ["COMPARE"].split(",")
    for spin in (2, 0):
        x, y = np.load(f"gpurun_out/ableg_{a}_{spin}.npy"), np.load(f"gpurun_out/ableg_{b}_{spin}.npy")
        print(f"spin {spin}: max |{a} - {b}| / max |{a}| = {np.abs(x - y).max() / np.abs(x).max():.3e}  (max |.| {np.abs(x).max():.3e}, {x.size} values)")
        r = np.abs(x - y)[-3145:] / np.abs(x).max()   # the row m = 3000 of the last component, l = 3000 ...
        k = int(np.argmax(r))
        print(f"   row m = 3000: max at l = {3000 + k}: {r[k]:.3e}; per 400 l: " + " ".join(f"{r[i:i + 400].max():.1e}" for i in range(0, 3145, 400)))
    sys.exit(0)
import torch
import heracles_amd as hx
hx.init(0)
tag = os.environ.get("TAG", "default")
nside, lmax = int(os.environ.get("NSIDE", 4096)), int(os.environ.get("LMAX", 6144))
plan = hx.Plan(nside, lmax)
for spin, ncomp in ((2, 20), (0, 10)):
    g = torch.Generator(device="cuda"); g.manual_seed(1234 + spin)
    m = torch.randn((ncomp, 12 * nside * nside), dtype=torch.float64, device="cuda", generator=g)
    for _ in range(2):
        alm = plan.map2alm(m, spin)
    res = []
    for rep in range(3):
        hx._lib.profile_enable(True); hx._lib.profile_reset()
        hx._lib.executed_flops(reset=True)
        alm = plan.map2alm(m, spin)
        torch.cuda.synchronize()
        ex = hx._lib.executed_flops(reset=True)
        res.append(round(hx._lib.profile_get("legendre_analysis")[1], 2))
        hx._lib.profile_enable(False)
    print(f"[{tag}] spin {spin} x {ncomp} comps: legendre_analysis ms {res}; executed flops: matrix {ex[0]:.4e}, recursion {ex[1]:.4e}", flush=True)
    bits = torch.view_as_real(alm).reshape(-1).view(torch.int64)
    print(f"[{tag}] spin {spin}: bit pattern of ALL {bits.numel()} doubles of the result: sum {int(bits.sum())} (mod 2^64), xor-fold {int(bits[::2].bitwise_xor(bits[1::2]).sum())}", flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    flat = alm.reshape(-1)
    # every 251st value plus whole orders near the poles' lead-in (m = 3000: the first 4000 values of its row)
    nlm = (lmax + 1) * (lmax + 2) // 2
    base = 3000 * (2 * lmax + 1 - 3000) // 2 + 3000
    samp = torch.cat([flat[::2503], alm[0, base:base + 3145], alm[ncomp - 1, base:base + 3145]])
    np.save(f"gpurun_out/ableg_{tag}_{spin}.npy", samp.cpu().numpy())
    del m, alm
    torch.cuda.empty_cache()
