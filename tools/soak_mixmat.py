"""Race screen of the mixing-matrix GEMM (k_mixmat_gemm_dma orders its LDS reads behind loads-to-LDS by hand): every build of the same
spectrum must be the same bit for bit, at sizes with full, partly filled and single rounds of tiles; HX_GEMM_DMA=0 in a second process
gives the register-staged kernel's result to compare with (the same products in the same order: equal bit for bit as well)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, heracles_amd as hx
hx.init(0)
reps = int(os.environ.get("REPS", 12))
for L in (95, 300, 1023, 2047, 3000, 4096, 6144):
    ell = np.arange(L + 1)
    wl = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / (0.08 * L * L)) + 1e-3 / (1.0 + ell) ** 2
    hs = set()
    for _ in range(reps):
        mm = hx.mixmat_eb(wl)
        hs.add(hashlib.sha1(np.ascontiguousarray(mm).tobytes()).hexdigest())
    print(f"L {L:5d}: {reps} builds, {len(hs)} distinct result(s)  sha1 {sorted(hs)[0][:16]}", flush=True)
    assert len(hs) == 1
