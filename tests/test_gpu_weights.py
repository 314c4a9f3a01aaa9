"""healpy pixel-weight files (hp.map2alm(use_pixel_weights=True, datapath=...), heracles/healpy.py:183-189): wire format and
GPU expansion of the compressed half-quadrant weights against the numpy restatement, the symmetries the compression rests on,
and the mapper picking the file up from its data path.  The weight VALUES are healpy data (absent): "parity unpinned"; the
synthetic file below stands in for them."""

import warnings

import numpy as np
import pytest


pytestmark = pytest.mark.gpu


def close(a, b, tol):
    scale = max(np.abs(b).max(), 1e-300)
    err = np.abs(np.asarray(a) - np.asarray(b)).max()
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("nside", [1, 2, 4, 16, 64, 6])
def test_expansion_matches_restatement_and_is_8fold_symmetric(oracle, nside, tmp_path):
    import heracles_amd as hx
    from heracles_amd import weights as hw

    rng = np.random.default_rng(nside)
    n = hw.compressed_size(nside)
    assert n == hx._lib.load().hx_pixel_weights_size(nside)
    comp = 1e-3 * rng.standard_normal(n)
    path = tmp_path / "full_weights" / hw.weights_filename(nside)
    path.parent.mkdir()
    hw.write_compressed_weights(path, nside, comp)
    assert hw.find_weights_file(tmp_path, nside) == str(path)
    back = hw.read_compressed_weights(path)
    np.testing.assert_array_equal(back, comp)
    full = hw.expand_pixel_weights(nside, back)
    np.testing.assert_array_equal(full, oracle.expand_full_weights(nside, comp))
    dev = hw.load_pixel_weights(tmp_path, nside)
    assert dev.is_cuda
    np.testing.assert_array_equal(dev.cpu().numpy(), full)
    # symmetries: under a rotation by 90 degrees about the axis, the mirror phi -> -phi and north <-> south a pixel centre maps
    # to a pixel centre, and the weight must agree there
    theta, phi = oracle.pix2ang(nside)
    lat = 90.0 - np.degrees(theta)
    for lon2, lat2 in ((np.degrees(phi) + 90.0, lat), (-np.degrees(phi), lat), (np.degrees(phi), -lat)):
        ip = oracle.ang2pix_ring(nside, np.mod(lon2, 360.0), lat2)
        np.testing.assert_array_equal(full[ip], full)
    with pytest.raises(hx.HxError):
        hw.expand_pixel_weights(nside, comp[:-1] if n > 1 else np.zeros(5))


def test_mapper_reads_weights_from_datapath(oracle, tmp_path):
    import heracles_amd as hx
    from heracles_amd import weights as hw

    nside, lmax = 16, 24
    rng = np.random.default_rng(9)
    comp = 1e-2 * rng.standard_normal(hw.compressed_size(nside))
    (tmp_path / "full_weights").mkdir()
    hw.write_compressed_weights(tmp_path / "full_weights" / hw.weights_filename(nside), nside, comp)
    m = rng.standard_normal(12 * nside**2)
    mapper = hx.HipHealpixMapper(nside, lmax, deconvolve=False, niter=0, datapath=tmp_path)
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # no "unit quadrature weights" warning: the file was found
        alm = mapper.transform(m)
    ref = oracle.map2alm(m[None], nside, lmax, spin=0, pix_weights=oracle.expand_full_weights(nside, comp))[0]
    np.testing.assert_allclose(alm, ref, atol=1e-11 * np.abs(ref).max())
    # the class-level DATAPATH of the reference (heracles/healpy.py:73, set from the configuration at cli.py:536-538)
    hx.HipHealpixMapper.DATAPATH = str(tmp_path)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            alm2 = hx.HipHealpixMapper(nside, lmax, deconvolve=False, niter=0).transform(m)
    finally:
        hx.HipHealpixMapper.DATAPATH = None
    np.testing.assert_array_equal(alm2, alm)


@pytest.mark.parametrize("nside,lmax,spin,ncomp", [(32, 64, 0, 3), (64, 100, 2, 4), (128, 160, 0, 10), (24, 40, 2, 10)])
def test_symmetric_weights_take_the_short_path_and_agree(oracle, nside, lmax, spin, ncomp):
    """healpy's full weights repeat over the four quadrants of a ring and from north to south; the ring kernels then read one weight
    per pixel pair instead of eight (flag set per call by k_pixw_symmetry).  The same transform with an array that is symmetric,
    and with one whose symmetry a single pixel breaks (generic path), against the oracle."""
    import heracles_amd as hx
    from heracles_amd import weights as hw

    rng = np.random.default_rng(5 * nside + spin)
    comp = 1e-2 * rng.standard_normal(hw.compressed_size(nside))
    w = np.asarray(hw.expand_pixel_weights(nside, comp))
    maps = rng.standard_normal((ncomp, 12 * nside**2))
    plan = hx.get_plan(nside, lmax)
    close(plan.map2alm(maps, spin, pix_weights=w), oracle.map2alm(maps, nside, lmax, spin=spin, pix_weights=w), 1e-11)
    w2 = w.copy()
    w2[12 * nside**2 - 3] *= 1.5  # a southern pixel: only the north -> south comparison sees it
    close(plan.map2alm(maps, spin, pix_weights=w2), oracle.map2alm(maps, nside, lmax, spin=spin, pix_weights=w2), 1e-11)
    w3 = w.copy()
    w3[5] = np.nan  # NaN never compares equal: generic path, and the NaN reaches the output as it would in healpy
    assert not np.isfinite(plan.map2alm(maps[:1] if spin == 0 else maps[:2], spin, pix_weights=w3)).all()
