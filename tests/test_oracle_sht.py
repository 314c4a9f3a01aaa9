"""Pins the CPU oracle's SHT (the third-party healpy is absent: "parity unpinned" against
healpy itself) with first-principles known answers: scipy Y_lm, an explicit finite-sum
spin-weighted Y_lm, and brute-force quadrature sums.  CPU only."""

import numpy as np
import pytest
from scipy.special import sph_harm_y

from helpers import idx, random_alm, sYlm


@pytest.mark.parametrize("nside,lmax", [(4, 8), (8, 16)])
def test_ring_geometry(oracle, nside, lmax):
    npix = 12 * nside**2
    seen = np.zeros(npix, dtype=int)
    area = 0.0
    for ring in range(1, 4 * nside):
        sp, nphi, z, sth, phi0 = oracle.ring_info(nside, ring)
        seen[sp : sp + nphi] += 1
        assert abs(z * z + sth * sth - 1) < 1e-14
        sp2, nphi2, z2, _, phi02 = oracle.ring_info(nside, 4 * nside - ring)
        assert nphi2 == nphi and abs(z2 + z) < 1e-15 and phi02 == phi0
    assert (seen == 1).all()
    theta, phi = oracle.pix2ang(nside)
    assert np.all(np.diff(theta) >= -1e-15)
    # mean of z over pixels vanishes, mean of z^2 is 1/3 (equal-area pixels)
    assert abs(np.cos(theta).mean()) < 1e-14
    assert abs((np.cos(theta) ** 2).mean() - 1 / 3) < 0.04 / nside**2


def test_spin0_synthesis_is_ylm(oracle):
    nside, lmax = 8, 16
    theta, phi = oracle.pix2ang(nside)
    nlm = oracle.nlm(lmax)
    for l, m in [(0, 0), (1, 0), (1, 1), (2, 1), (5, 3), (16, 16), (16, 0), (9, 8)]:
        alm = np.zeros(nlm, complex)
        alm[idx(lmax, l, m)] = 1.0 + 0.5j if m > 0 else 1.0
        mp = oracle.alm2map(alm, nside, lmax)
        Y = sph_harm_y(l, m, theta, phi)
        ref = Y.real if m == 0 else 2 * np.real((1.0 + 0.5j) * Y)
        np.testing.assert_allclose(mp, ref, atol=1e-13)


@pytest.mark.parametrize("use_fft", [True, False])
def test_spin0_analysis_is_quadrature_sum(oracle, use_fft):
    nside, lmax = 8, 20  # lmax > 2*nside: exercises m aliasing on the polar rings
    rng = np.random.default_rng(1)
    theta, phi = oracle.pix2ang(nside)
    npix = 12 * nside**2
    mp = rng.standard_normal(npix)
    a = oracle.map2alm(mp, nside, lmax, use_fft=use_fft)
    for l, m in [(0, 0), (3, 2), (20, 20), (20, 0), (17, 16), (11, 5)]:
        ref = (np.conj(sph_harm_y(l, m, theta, phi)) * mp).sum() * 4 * np.pi / npix
        assert abs(a[idx(lmax, l, m)] - ref) < 1e-13


def test_spin2_matches_explicit_spin_weighted_harmonics(oracle):
    nside, lmax = 8, 12
    rng = np.random.default_rng(2)
    theta, phi = oracle.pix2ang(nside)
    E = random_alm(rng, lmax, 2)
    B = random_alm(rng, lmax, 2)
    # Q + iU = -sum_{lm} (E + iB) 2Y_lm (HEALPix / Zaldarriaga-Seljak convention)
    P = np.zeros(12 * nside**2, complex)
    for m in range(-lmax, lmax + 1):
        for l in range(max(abs(m), 2), lmax + 1):
            if m >= 0:
                e, b = E[idx(lmax, l, m)], B[idx(lmax, l, m)]
            else:
                e = (-1) ** m * np.conj(E[idx(lmax, l, -m)])
                b = (-1) ** m * np.conj(B[idx(lmax, l, -m)])
            P += -(e + 1j * b) * sYlm(2, l, m, theta, phi)
    QU = oracle.alm2map(np.stack([E, B]), nside, lmax, spin=2)
    np.testing.assert_allclose(QU[0], P.real, atol=2e-10)
    np.testing.assert_allclose(QU[1], P.imag, atol=2e-10)
    # analysis = quadrature sum against conj(sYlm)
    eb = oracle.map2alm(QU, nside, lmax, spin=2)
    dA = 4 * np.pi / (12 * nside**2)
    for l, m in [(2, 0), (2, 1), (2, 2), (5, 3), (7, 0), (12, 12), (12, 1), (10, 9)]:
        a2p = ((QU[0] + 1j * QU[1]) * np.conj(sYlm(2, l, m, theta, phi))).sum() * dA
        a2m = ((QU[0] - 1j * QU[1]) * np.conj(sYlm(-2, l, m, theta, phi))).sum() * dA
        assert abs(-(a2p + a2m) / 2 - eb[0][idx(lmax, l, m)]) < 1e-10
        assert abs(1j * (a2p - a2m) / 2 - eb[1][idx(lmax, l, m)]) < 1e-10


@pytest.mark.parametrize("nside,lmax", [(16, 16), (12, 12), (32, 24)])
@pytest.mark.parametrize("spin", [0, 2])
def test_roundtrip_band_limited_with_iterations(oracle, nside, lmax, spin):
    rng = np.random.default_rng(3)
    alm = random_alm(rng, lmax, spin, (2,))
    mp = oracle.alm2map(alm, nside, lmax, spin=spin)
    back0 = oracle.map2alm(mp, nside, lmax, spin=spin)
    back3 = oracle.map2alm(mp, nside, lmax, spin=spin, niter=3)
    e0, e3 = np.abs(back0 - alm).max(), np.abs(back3 - alm).max()
    assert e0 < 0.05 and e3 < e0 * 0.1


def test_fft_equals_direct_dft(oracle):
    rng = np.random.default_rng(4)
    nside, lmax = 6, 14  # non power-of-two nside: Bluestein on every ring
    mp = rng.standard_normal((2, 12 * nside**2))
    a = oracle.map2alm(mp, nside, lmax, spin=2, use_fft=True)
    b = oracle.map2alm(mp, nside, lmax, spin=2, use_fft=False)
    np.testing.assert_allclose(a, b, atol=1e-14)
    alm = random_alm(rng, lmax, 0)
    np.testing.assert_allclose(
        oracle.alm2map(alm, nside, lmax, use_fft=True), oracle.alm2map(alm, nside, lmax, use_fft=False), atol=1e-13
    )


def test_parseval(oracle):
    rng = np.random.default_rng(5)
    nside, lmax = 32, 40
    alm = random_alm(rng, lmax, 0)
    mp = oracle.alm2map(alm, nside, lmax)
    cl = oracle.alm2cl(alm)
    lhs = ((2 * np.arange(lmax + 1) + 1) * cl).sum() / (4 * np.pi)
    assert abs(lhs - (mp**2).mean()) < 2e-3 * lhs


def test_extended_precision_lambda_helper():
    """tests/helpers.lambda_lm_column (the closed form of the full-size GPU test) against scipy."""
    import scipy.special as sp
    from helpers import lambda_lm_column

    th = np.array([0.01, 0.7, 1.5, 3.0])
    for m in (0, 1, 5, 30):
        lam = lambda_lm_column(m, 60, np.cos(th), np.sin(th)).astype(float)
        ref = np.array([[sp.sph_harm_y(l, m, t, 0.0).real for t in th] for l in range(m, 61)])
        assert np.abs(lam - ref).max() <= 1e-13 * np.abs(ref).max()
    # far below the double range (sin^m(theta) ~ 1e-1400) the seed is still a normal number
    lam = lambda_lm_column(700, 800, np.cos([1e-2]), np.sin([1e-2]))
    assert np.all(np.isfinite(lam)) and lam[0, 0] != 0 and float(lam[0, 0]) == 0.0
