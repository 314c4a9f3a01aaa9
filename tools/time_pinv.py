"""hx_pinv (blocked one-sided Jacobi SVD) on mixing-matrix-like band matrices: seconds and sweeps per size, and the error against
np.linalg.pinv where numpy finishes quickly (N <= NREF).  SIZES / NREF from the environment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, heracles_amd as hx
from heracles_amd.twopoint import pinv
hx.init(0)
sizes = [int(x) for x in os.environ.get("SIZES", "1025,2049,4097,6145").split(",")]
nref = int(os.environ.get("NREF", 2049))
rng = np.random.default_rng(1)
for n in sizes:
    i = torch.arange(n, dtype=torch.float64, device="cuda")
    a = torch.exp(-0.5 * ((i[:, None] - i[None, :]) / 4.0) ** 2) + 1e-3 * torch.randn((n, n), dtype=torch.float64, device="cuda")
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        out, info = pinv(a, 1e-5, device="cuda", info=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    msg = f"n {n}: {dt:.3f} s, {info['sweeps']} sweeps, kept {info['kept']}, sigma {info['largest']:.3g} .. {info['smallest_kept']:.3g}"
    chk = (out @ a @ out - out).abs().max().item() / out.abs().max().item()   # Penrose condition X A X = X
    msg += f", |X A X - X| / |X| = {chk:.1e}"
    if n <= nref:
        t = time.perf_counter(); ref = np.linalg.pinv(a.cpu().numpy(), rcond=1e-5); tn = time.perf_counter() - t
        msg += f", vs numpy ({tn:.2f} s): {np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max():.1e}"
    print(msg, flush=True)
