"""GPU parity of the point transform (hx_pointsht_adjoint, heracles/ducc.py:92-133) against the direct-sum oracle.
Floating point: the reference asks ducc0 for epsilon = 1e-12 (float64 values) / 1e-5 (float32); the tests allow
1e-11 / 1e-5 of the largest |alm| (the error of a non-uniform FFT scales with sum |v_p|, not with the single alm)."""
import numpy as np
import pytest

from oracle import hxoracle as oracle

pytestmark = pytest.mark.gpu


def _points(rng, n):
    theta = np.arccos(rng.uniform(-1, 1, n))
    phi = rng.uniform(0, 2 * np.pi, n)
    return theta, phi


def _err(got, want):
    return np.abs(got - want).max() / np.abs(want).max()


@pytest.mark.parametrize("lmax", [0, 1, 7, 31, 48, 100])  # 31: smallest oversampling (n1 / (2 lmax + 1) = 2.03)
@pytest.mark.parametrize("spin", [0, 2])
def test_random_points_against_direct_sum(lmax, spin):
    import heracles_amd as hx

    rng = np.random.default_rng(100 * lmax + spin)
    n = 500
    theta, phi = _points(rng, n)
    v = rng.normal(size=(4, n))
    got = hx.PointSHT(lmax).adjoint_synthesis(np.stack([theta, phi], axis=1), v, spin=spin)
    want = oracle.points2alm(theta, phi, v, lmax, spin=spin)
    assert got.shape == want.shape
    if spin == 2 and lmax < 2:
        assert np.abs(got).max() == 0.0
        return
    assert _err(got, want) < 1e-11


def test_poles_seam_and_longitude_range():
    import heracles_amd as hx

    lmax = 40
    theta = np.array([0.0, np.pi, 1e-9, np.pi - 1e-9, 0.7, 0.7, 2.0, 2.0, np.pi / 2])
    phi = np.array([0.3, 1.0, 0.0, 6.0, 0.0, 2 * np.pi - 1e-12, -1.0, 7.5, 4 * np.pi + 0.25])
    v = np.arange(1.0, 2 * theta.size + 1).reshape(2, -1)
    sht = hx.PointSHT(lmax)
    for spin in (0, 2):
        got = sht.adjoint_synthesis(np.stack([theta, phi], axis=1), v, spin=spin)
        want = oracle.points2alm(theta, phi, v, lmax, spin=spin)
        assert _err(got, want) < 1e-11


def test_many_components_and_device_tensors():
    import torch
    import heracles_amd as hx

    rng = np.random.default_rng(9)
    lmax, n = 24, 300
    theta, phi = _points(rng, n)
    v = rng.normal(size=(20, n))  # more than one sweep of the Legendre kernel
    loc = np.stack([theta, phi], axis=1)
    sht = hx.PointSHT(lmax)
    want = oracle.points2alm(theta, phi, v, lmax, spin=0)
    got = sht.adjoint_synthesis(torch.as_tensor(loc).cuda(), torch.as_tensor(v).cuda(), spin=0)
    assert got.is_cuda and _err(got.cpu().numpy(), want) < 1e-11
    want2 = oracle.points2alm(theta, phi, v, lmax, spin=2)
    assert _err(sht.adjoint_synthesis(loc, v, spin=2), want2) < 1e-11


def test_float32_accuracy_class():
    import heracles_amd as hx

    rng = np.random.default_rng(3)
    lmax, n = 64, 400
    theta, phi = _points(rng, n)
    v = rng.normal(size=(1, n))
    sht = hx.PointSHT(lmax, epsilon=1e-5)
    assert sht.kernel_width < hx.PointSHT(lmax).kernel_width
    got = sht.adjoint_synthesis(np.stack([theta, phi], axis=1), v)
    assert _err(got, oracle.points2alm(theta, phi, v, lmax)) < 1e-5


@pytest.mark.parametrize("spin", [0, 2])
def test_pixel_centres_reproduce_map2alm(spin):
    """A HEALPix map is a set of points of weight 4 pi / npix: both device paths must agree."""
    import heracles_amd as hx

    nside, lmax = 32, 64
    npix = 12 * nside * nside
    rng = np.random.default_rng(11 + spin)
    maps = rng.normal(size=(2, npix))
    theta, phi = oracle.pix2ang(nside)
    want = hx.get_plan(nside, lmax).map2alm(maps, spin)
    got = hx.PointSHT(lmax).adjoint_synthesis(np.stack([theta, phi], axis=1), maps * (4 * np.pi / npix), spin=spin)
    assert _err(got, np.asarray(want)) < 1e-11


@pytest.mark.parametrize("lmax", [2100, 4200])  # FFT lengths 16384 and 32768: radix-2 / radix-4 step over in-LDS transforms
def test_long_transforms_on_sampled_m(lmax):
    """Away from the poles: 1e-11.  A point within a few rings of a pole puts all its weight on rings whose lambda_lm is
    evaluated through x = cos(theta) in float64: a three-term recursion then carries l * 1.1e-16 / sin(theta) (the ring
    is displaced by < 1e-7 arc seconds) -- the same conditioning as on the first HEALPix rings, independent of the NUFFT;
    the oracle runs its recursion in extended precision.  Bound used: 10 * lmax * 1.1e-16 / sin(first ring)."""
    import heracles_amd as hx

    rng = np.random.default_rng(lmax)
    n = 40
    sht = hx.PointSHT(lmax)
    stride = 97

    def worst_error(theta, phi, v):
        got = sht.adjoint_synthesis(np.stack([theta, phi], axis=1), v, spin=2)
        oracle.set_mstride(stride)
        try:
            want = oracle.points2alm(theta, phi, v, lmax, spin=2)
        finally:
            oracle.set_mstride(1)
        worst = 0.0
        for m in range(0, lmax + 1, stride):
            lo = m * (2 * lmax + 1 - m) // 2 + m
            hi = lo + lmax - m + 1
            worst = max(worst, np.abs(got[:, lo:hi] - want[:, lo:hi]).max())
        return worst / np.abs(want).max()

    theta = np.arccos(rng.uniform(-0.995, 0.995, n))
    phi = rng.uniform(0, 2 * np.pi, n)
    v = rng.normal(size=(2, n))
    assert worst_error(theta, phi, v) < 1e-11
    theta[:3] = [1e-4, np.pi - 3e-4, np.pi / 2]
    assert worst_error(theta, phi, v) < 10 * lmax * 1.1e-16 / np.sin(np.pi / sht.nrings_circle)


def test_crowded_patch_many_points():
    """2e5 points, half of them inside one square degree: many hardware atomics on the same grid cells from all
    compute dies (the sums must not lose updates), the rest anywhere."""
    import heracles_amd as hx

    rng = np.random.default_rng(77)
    lmax, n = 48, 200_000
    theta, phi = _points(rng, n)
    theta[: n // 2] = 1.0 + rng.uniform(0, np.radians(1.0), n // 2)
    phi[: n // 2] = 2.0 + rng.uniform(0, np.radians(1.0), n // 2)
    v = rng.uniform(0.5, 1.5, size=(1, n))  # no cancellation: a lost update would show
    got = hx.PointSHT(lmax).adjoint_synthesis(np.stack([theta, phi], axis=1), v)
    want = oracle.points2alm(theta, phi, v, lmax)
    assert _err(got, want) < 1e-11


@pytest.mark.parametrize("lmax", [5, 48, 300])  # grids smaller than, comparable to and larger than one 64-cell tile
def test_tiled_spreading_path(lmax, monkeypatch):
    """Large catalogues are spread through LDS tiles (one sort per call); HX_NUFFT_TILES=1 forces that path here."""
    import heracles_amd as hx

    monkeypatch.setenv("HX_NUFFT_TILES", "1")
    rng = np.random.default_rng(lmax)
    n = 3000
    theta, phi = _points(rng, n)
    theta[:4] = [0.0, np.pi, 1e-7, np.pi - 1e-7]
    phi[:4] = [0.0, 2 * np.pi - 1e-13, 6.28, 1e-9]
    v = rng.normal(size=(4, n))
    v[0, :10] = 0.0
    loc = np.stack([theta, phi], axis=1)
    sht = hx.PointSHT(lmax)
    tol = 1e-11 if lmax < 100 else 10 * lmax * 1.1e-16 / np.sin(np.pi / sht.nrings_circle)
    for spin in (0, 2):
        assert _err(sht.adjoint_synthesis(loc, v, spin=spin), oracle.points2alm(theta, phi, v, lmax, spin=spin)) < tol
    loc[7, 0] = -0.1
    with pytest.raises(ValueError):
        sht.adjoint_synthesis(loc, v)


def test_invalid_points_raise():
    import heracles_amd as hx

    sht = hx.PointSHT(16)
    loc = np.array([[0.5, 1.0], [3.5, 1.0]])
    with pytest.raises(ValueError):
        sht.adjoint_synthesis(loc, np.ones((1, 2)))
    loc[1, 0] = np.nan
    with pytest.raises(ValueError):
        sht.adjoint_synthesis(loc, np.ones((1, 2)))
    with pytest.raises(ValueError):
        sht.adjoint_synthesis(loc[:1], np.ones((3, 1)), spin=2)


def test_discrete_mapper_map_values():
    """DiscreteMapper.map_values semantics (heracles/ducc.py:92-133): lon / lat in degrees, 1-D values flattened,
    the result is ADDED to data."""
    from heracles_amd import HipDiscreteMapper

    rng = np.random.default_rng(21)
    lmax, n = 30, 200
    lon = rng.uniform(-180, 540, n)
    lat = np.degrees(np.arcsin(rng.uniform(-1, 1, n)))
    theta, phi = np.radians(90 - lat), np.radians(lon % 360)
    mapper = HipDiscreteMapper(lmax)
    data = mapper.create(spin=0)
    data += 1.0
    w = rng.normal(size=n)
    mapper.map_values(lon, lat, data, w)
    want = 1.0 + oracle.points2alm(theta, phi, w[None], lmax)[0]
    assert _err(data, want) < 1e-11
    assert data.dtype.metadata["geometry"] == "discrete"
    data2 = mapper.create(2, spin=2)
    g = rng.normal(size=(2, n))
    mapper.map_values(lon, lat, data2, g, spin=2)
    mapper.map_values(lon, lat, data2, g.astype(np.float32), spin=2)
    want2 = oracle.points2alm(theta, phi, g, lmax, spin=2)
    assert _err(data2, 2 * want2) < 1e-5
