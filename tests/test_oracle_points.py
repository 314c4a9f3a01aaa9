"""The direct-sum oracle of the point transform (oracle.points2alm, heracles/ducc.py:121-128) against independent
evaluations: scipy's spherical harmonics (spin 0), the HEALPix ring restatement on pixel centres (both spins: a map is
a set of points of weight 4 pi / npix), and closed forms at the poles.  The reference holds no fixture for
DiscreteMapper.map_values (tests/test_ducc.py covers resample only): parity of this function rests on the definition."""
import numpy as np
import pytest

from oracle import hxoracle as oracle


def _idx(lmax, l, m):
    return m * (2 * lmax + 1 - m) // 2 + l


def test_spin0_against_scipy():
    sp = pytest.importorskip("scipy.special")
    ylm = getattr(sp, "sph_harm_y", None)
    rng = np.random.default_rng(5)
    lmax, n = 20, 150
    theta = np.arccos(rng.uniform(-1, 1, n))
    phi = rng.uniform(0, 2 * np.pi, n)
    v = rng.normal(size=(2, n))
    got = oracle.points2alm(theta, phi, v, lmax, spin=0)
    for m in range(lmax + 1):
        for l in range(m, lmax + 1):
            y = ylm(l, m, theta, phi) if ylm is not None else sp.sph_harm(m, l, phi, theta)
            want = (v * np.conj(y)[None, :]).sum(axis=1)
            np.testing.assert_allclose(got[:, _idx(lmax, l, m)], want, rtol=0, atol=2e-12 * np.abs(v).sum())


@pytest.mark.parametrize("spin", [0, 2])
def test_pixel_centres_reproduce_map2alm(spin):
    nside, lmax = 8, 16
    npix = 12 * nside * nside
    rng = np.random.default_rng(spin + 1)
    maps = rng.normal(size=(2, npix))
    theta, phi = oracle.pix2ang(nside)
    want = oracle.map2alm(maps, nside, lmax, spin=spin)
    got = oracle.points2alm(theta, phi, maps * (4 * np.pi / npix), lmax, spin=spin)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-13 * np.abs(want).max() * 50)


def test_point_at_the_pole():
    # Y_lm(0, phi) = sqrt((2l+1)/(4 pi)) delta_m0;  spin 2 at the pole: only m = 2 survives
    lmax = 12
    a = oracle.points2alm([0.0], [0.3], [[2.5]], lmax, spin=0)[0]
    for l in range(lmax + 1):
        assert a[_idx(lmax, l, 0)] == pytest.approx(2.5 * np.sqrt((2 * l + 1) / (4 * np.pi)), rel=1e-14)
    assert np.abs(a[lmax + 1:]).max() == 0.0
    e = oracle.points2alm([0.0], [0.0], [[1.0], [0.0]], lmax, spin=2)
    nz = np.zeros(e.shape[1], dtype=bool)
    nz[[_idx(lmax, l, 2) for l in range(2, lmax + 1)]] = True
    assert np.abs(e[:, ~nz]).max() < 1e-300
    assert np.abs(e[0, nz]).min() > 0.1
