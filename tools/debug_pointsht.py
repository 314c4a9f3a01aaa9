"""Where the error of the point transform sits (per sampled m), against the direct-sum oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import heracles_amd as hx
from oracle import hxoracle as oracle
lmax = int(os.environ.get("LMAX", 4200)); spin = int(os.environ.get("SPIN", 0)); stride = int(os.environ.get("STRIDE", 211))
rng = np.random.default_rng(1)
n = int(os.environ.get("NPTS", 20))
theta = np.arccos(rng.uniform(-1, 1, n)); phi = rng.uniform(0, 2 * np.pi, n)
if os.environ.get("POLE"): theta[0] = 1e-4
v = rng.normal(size=(2, n))
sht = hx.PointSHT(lmax)
print("N", sht.nrings_circle, "n1", sht.ngrid, "W", sht.kernel_width)
got = sht.adjoint_synthesis(np.stack([theta, phi], axis=1), v, spin=spin)
oracle.set_mstride(stride)
want = oracle.points2alm(theta, phi, v, lmax, spin=spin)
sc = np.abs(want).max()
for m in range(0, lmax + 1, stride):
    lo = m * (2 * lmax + 1 - m) // 2 + m; hi = lo + lmax - m + 1
    d = np.abs(got[:, lo:hi] - want[:, lo:hi])
    i = np.unravel_index(d.argmax(), d.shape)
    print(f"m {m:5d} max err {d.max()/sc:.2e} at l = {m + i[1]}", flush=True)
