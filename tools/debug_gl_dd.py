"""GPU box: the double-double Gauss-Legendre nodes of libhxsht against a long-double Newton iteration, and what they do to one
high-l element of the (0,0) mixing matrix (exact value from big-integer 3j sums, tests/test_oracle_golden.py)."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import heracles_amd as hx
from heracles_amd import _lib

hx.init(0)
L = 4096
n = 3 * L // 2 + 1
x, w, xlo = np.empty(n), np.empty(n), np.empty(n)
_lib.check(_lib.load().hx_gauss_legendre_dd(n, _lib.ptr(x), _lib.ptr(w), _lib.ptr(xlo)))
ld = np.longdouble
xl = x.astype(ld)
for it in range(2):
    p0, p1 = np.ones_like(xl), xl.copy()
    for k in range(1, n):
        p0, p1 = p1, ((2 * k + 1) * xl * p1 - k * p0) / (k + 1)
    dp = n * (xl * p1 - p0) / (xl * xl - 1)
    xl = xl - p1 / dp
ref = (xl - x.astype(ld)).astype(np.float64)
print("max |x - root| (double part):", np.abs(ref).max(), " max |xlo - ref|:", np.abs(xlo - ref).max(), " max |xlo|:", np.abs(xlo).max())
k = np.argsort(-np.abs(xlo - ref))[:5]
print("worst nodes:", k, x[k], xlo[k], ref[k])
ell = np.arange(L + 1)
W = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / 3000.0)
m = hx.mixmat(W, spin=(0, 0))
print("M[L, L] - exact:", m[L, L] - 10.817601351044953)
