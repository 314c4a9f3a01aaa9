"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (host build only: GPU sanitizers are not available on
the pool).  `make -C oracle asan` builds libhxoracle_asan.so; a child interpreter with libasan preloaded runs every oracle entry
point on small inputs (odd sizes, lmax > 2 nside, spin 0 and 2, iterations, unequal lmax) and must exit cleanly."""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r"""
import numpy as np
from oracle import hxoracle as ho
rng = np.random.default_rng(5)
for nside, lmax in ((4, 9), (8, 12), (5, 7)):
    npix = 12 * nside * nside
    t = rng.standard_normal((2, npix)); qu = rng.standard_normal((2, npix))
    a0 = ho.map2alm(t, nside, lmax, spin=0, niter=1)
    a2 = ho.map2alm(qu, nside, lmax, spin=2, pix_weights=np.ones(npix), ring_weights=np.ones(2 * nside))
    m0 = ho.alm2map(a0, nside, lmax); m2 = ho.alm2map(a2, nside, lmax, spin=2)
    assert np.isfinite(m0).all() and np.isfinite(m2).all()
    ho.alm2cl(a0[0], a2); ho.alm2cl(a2, a2, lmax=lmax - 2)
th = rng.uniform(0.05, 3.0, 37); ph = rng.uniform(0, 6.28, 37)
ho.points2alm(th, ph, rng.standard_normal((1, 37)), 11, spin=0)
ho.points2alm(th, ph, rng.standard_normal((2, 37)), 11, spin=2)
b = rng.standard_normal((11 * 12 // 2, 2)) @ [1, 1j]; c = rng.standard_normal((21 * 22 // 2, 2)) @ [1, 1j]
ho.alm2cl(b, c); ho.alm2cl(c, b, lmax=20)
assert ho.alm2lmax(b.size) == 10
x, w = ho.gauss_legendre(9)
for ab in ((0, 0), (2, 0), (2, 2), (2, -2), (1, 1), (-1, 1)):
    ho.wigner_d(13, ab[0], ab[1], 0.3)
ho.legendre_funcs(17, 0.9991); ho.legendre_funcs(17, -0.4)
cl4 = rng.standard_normal((14, 4))
ho.corr2cl(ho.cl2corr(cl4))
wl = 1.0 / (1.0 + np.arange(9)) ** 2
ho.mixmat(wl); ho.mixmat(wl, l1max=5, l2max=7, spin=(0, 2)); ho.mixmat_eb(wl, l1max=6, l2max=6)
print("sanitized oracle ok")
"""


def test_oracle_under_asan_ubsan():
    try:
        libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    except (OSError, subprocess.CalledProcessError):
        pytest.skip("gcc not available")
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not installed with this gcc")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               HXORACLE_LIB=os.path.join(ROOT, "oracle", "libhxoracle_asan.so"), PYTHONPATH=ROOT, OMP_NUM_THREADS="2")
    res = subprocess.run([sys.executable, "-c", DRIVER], env=env, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert res.returncode == 0 and "sanitized oracle ok" in res.stdout, (res.stdout[-2000:], res.stderr[-4000:])
