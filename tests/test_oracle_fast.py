"""The vectorised CPU restatement of map2alm that bench.py's ``cpu_baseline`` leg times (oracle/hx_cpu_fast.c) is CHECKED BY the
scalar oracle -- never the other way round: spin 0 and 2, with and without pixel weights, shapes that exercise the Bluestein
rings, the power-of-two belt, ring padding (nside < 4), the pruned polar rings and the scaled start of the recursions (lmax up
to 3 nside / 2 ... 3 nside).  Tolerance 1e-11 of the largest alm (measured 1e-15 ... 2e-14)."""

import numpy as np
import pytest

from oracle import hxfast


@pytest.mark.skipif(not hxfast.supported(), reason="no AVX-512 on this CPU")
@pytest.mark.parametrize("nside,lmax", [(1, 2), (2, 5), (4, 8), (8, 23), (16, 40), (32, 95), (64, 128), (128, 300)])
def test_fast_map2alm_against_the_oracle(oracle, nside, lmax):
    rng = np.random.default_rng(nside + lmax)
    npix = 12 * nside * nside
    for spin in (0, 2):
        x = rng.standard_normal((2, npix))
        for pw in (None, 1.0 + 0.01 * rng.standard_normal(npix)):
            ref = oracle.map2alm(x, nside, lmax, spin=spin, pix_weights=pw)
            got, _ = hxfast.map2alm(x, nside, lmax, spin=spin, pix_weights=pw)
            assert got.shape == ref.shape
            assert np.abs(got - ref).max() <= 1e-11 * np.abs(ref).max(), (spin, pw is None)


@pytest.mark.skipif(not hxfast.supported(), reason="no AVX-512 on this CPU")
def test_fast_map2alm_deep_scaling(oracle):
    """nside 512 / lmax 768 at high m only: the seeds of the polar rings start hundreds of binary orders below 1 (several rescalings)"""
    nside, lmax = 256, 700
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 12 * nside * nside))
    oracle.set_mstride(50)
    try:
        for spin in (0, 2):
            ref = oracle.map2alm(x, nside, lmax, spin=spin)
            got, _ = hxfast.map2alm(x, nside, lmax, spin=spin)
            scale = np.abs(got).max()
            for m in range(0, lmax + 1, 50):
                b = m * (2 * lmax + 1 - m) // 2
                assert np.abs(got[:, b + m : b + lmax + 1] - ref[:, b + m : b + lmax + 1]).max() <= 1e-11 * scale, (spin, m)
    finally:
        oracle.set_mstride(1)


@pytest.mark.skipif(not hxfast.supported(), reason="no AVX-512 on this CPU")
def test_fast_alm2cl_against_the_oracle(oracle):
    rng = np.random.default_rng(9)
    lmax = 97
    n = (lmax + 1) * (lmax + 2) // 2
    a = rng.standard_normal((n, 2)) @ [1, 1j]
    b = rng.standard_normal((n, 2)) @ [1, 1j]
    np.testing.assert_allclose(hxfast.alm2cl(a, b), oracle.alm2cl(a, b), rtol=1e-12, atol=1e-15)
