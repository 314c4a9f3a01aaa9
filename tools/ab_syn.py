"""A/B of synthesis builds on one device at the bench size (the counterpart of tools/ab_leg.py): Legendre part of alm2map for ten spin-2
fields and ten spin-0 maps, and a strided sample of the maps in gpurun_out/absyn_<TAG>_<spin>.npy (COMPARE=tagA,tagB prints the difference)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("COMPARE"):
    a, b = os.environ["COMPARE"].split(",")
    for spin in (2, 0):
        x, y = np.load(f"gpurun_out/absyn_{a}_{spin}.npy"), np.load(f"gpurun_out/absyn_{b}_{spin}.npy")
        print(f"spin {spin}: max |{a} - {b}| / max |{a}| = {np.abs(x - y).max() / np.abs(x).max():.3e}  (max |.| {np.abs(x).max():.3e}, {x.size} values)")
    sys.exit(0)
import torch
import heracles_amd as hx
hx.init(0)
tag = os.environ.get("TAG", "default")
nside, lmax = int(os.environ.get("NSIDE", 4096)), int(os.environ.get("LMAX", 6144))
plan = hx.Plan(nside, lmax)
nlm = (lmax + 1) * (lmax + 2) // 2
for spin, ncomp in ((2, 20), (0, 10)):
    g = torch.Generator(device="cuda"); g.manual_seed(4321 + spin)
    alm = torch.view_as_complex(torch.randn((ncomp, nlm, 2), dtype=torch.float64, device="cuda", generator=g))
    out = torch.empty((ncomp, 12 * nside * nside), dtype=torch.float64, device="cuda")
    plan.alm2map(alm, spin, out=out)
    res = []
    for rep in range(3):
        hx._lib.profile_enable(True); hx._lib.profile_reset()
        plan.alm2map(alm, spin, out=out)
        torch.cuda.synchronize()
        res.append(round(hx._lib.profile_get("legendre_synthesis")[1], 2))
        hx._lib.profile_enable(False)
    print(f"[{tag}] alm2map spin {spin} x {ncomp} comps: legendre_synthesis ms {res}", flush=True)
    bits = out.reshape(-1).view(torch.int64)
    print(f"[{tag}] spin {spin}: bit pattern of ALL {bits.numel()} doubles of the result: sum {int(bits.sum())} (mod 2^64), xor-fold {int(bits[::2].bitwise_xor(bits[1::2]).sum())}", flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    # every 4099th pixel of every map plus the first 200000 pixels (the polar cap, where the lead-in of the chains is longest)
    samp = torch.cat([out[:, ::4099].reshape(-1), out[0, :200000], out[ncomp - 1, :200000]])
    np.save(f"gpurun_out/absyn_{tag}_{spin}.npy", samp.cpu().numpy())
    del alm, out
    torch.cuda.empty_cache()
