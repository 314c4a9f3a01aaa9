"""One parity case per configuration of BASELINE.json ("configs"), through the drop-in surface
(HipHealpixMapper.transform / angular_power_spectra / mixmat_eb) against the oracle.

  0  1 spin-0 map, nside 64, lmax 128: auto-Cl
  1  1 spin-0 + 1 spin-2 map, nside 1024, lmax 2048: auto + cross Cl          (oracle in full)
  2  10 bins x (spin 0, spin 2), nside 2048, lmax 3072: one bin here, oracle on every 96th m;
     the all-pairs sharding over ranks is covered on CPU (tests/test_distributed_cpu.py)
  3  mixing matrix at lmax 4096: identities that hold at any size + the oracle on a corner
  4  nside 4096, lmax 6144: tests/test_gpu_fullsize.py
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fields_spectra(hx, oracle, nside, lmax, seed):
    """maps -> alms through the mapper (niter = 0, no deconvolution), all spectra on the GPU, the same
    through the oracle."""
    rng = np.random.default_rng(seed)
    npix = 12 * nside * nside
    t = rng.standard_normal(npix)
    qu = rng.standard_normal((2, npix))
    mapper = hx.HipHealpixMapper(nside, lmax, deconvolve=False, niter=0)
    hx.update_metadata(t, spin=0, nside=nside)
    hx.update_metadata(qu, spin=2, nside=nside)
    alms = {("POS", 0): mapper.transform(t, spin=0), ("SHE", 0): mapper.transform(qu, spin=2)}
    cls = hx.angular_power_spectra(alms, debias=False)
    a0 = oracle.map2alm(t[None], nside, lmax, spin=0)[0]
    a2 = oracle.map2alm(qu, nside, lmax, spin=2)
    return alms, cls, a0, a2


def test_config0_nside64_lmax128(oracle):
    import heracles_amd as hx

    alms, cls, a0, _ = _fields_spectra(hx, oracle, 64, 128, 50)
    scale = np.abs(a0).max()
    assert np.abs(alms[("POS", 0)] - a0).max() <= 1e-12 * scale
    np.testing.assert_allclose(np.asarray(cls[("POS", "POS", 0, 0)]), oracle.alm2cl(a0, a0), rtol=1e-11, atol=1e-16)


def test_config1_nside1024_lmax2048(oracle):
    import heracles_amd as hx

    alms, cls, a0, a2 = _fields_spectra(hx, oracle, 1024, 2048, 51)
    assert np.abs(alms[("POS", 0)] - a0).max() <= 1e-11 * np.abs(a0).max()
    assert np.abs(alms[("SHE", 0)] - a2).max() <= 1e-11 * np.abs(a2).max()
    assert list(cls.keys()) == [("POS", "POS", 0, 0), ("POS", "SHE", 0, 0), ("SHE", "SHE", 0, 0)]
    for key, ref in ((("POS", "POS", 0, 0), oracle.alm2cl(a0, a0)), (("POS", "SHE", 0, 0), oracle.alm2cl(a0, a2)),
                     (("SHE", "SHE", 0, 0), oracle.alm2cl(a2, a2))):
        got = np.asarray(cls[key])
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-13 * np.abs(ref).max())


def test_config2_nside2048_lmax3072_sampled_m(oracle):
    import heracles_amd as hx

    nside, lmax, stride = 2048, 3072, 96
    rng = np.random.default_rng(52)
    maps = rng.standard_normal((3, 12 * nside * nside))
    plan = hx.get_plan(nside, lmax)
    got0 = plan.map2alm(maps[:1], 0)
    got2 = plan.map2alm(maps[1:], 2)
    oracle.set_mstride(stride)
    try:
        ref0 = oracle.map2alm(maps[:1], nside, lmax, spin=0)
        ref2 = oracle.map2alm(maps[1:], nside, lmax, spin=2)
    finally:
        oracle.set_mstride(1)
    for got, ref in ((got0, ref0), (got2, ref2)):
        scale = np.abs(got).max()
        for m in range(0, lmax + 1, stride):
            base = m * (2 * lmax + 1 - m) // 2
            sl = slice(base + m, base + lmax + 1)
            assert np.abs(got[:, sl] - ref[:, sl]).max() <= 1e-10 * scale, m


def test_config3_mixing_matrix_lmax4096(oracle):
    import heracles_amd as hx

    L = 4096
    ell = np.arange(L + 1)
    wl = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / 3000.0) + 1e-3 / (1.0 + ell) ** 2
    eb = hx.mixmat_eb(wl)                       # l1max = l2max = l3max = 4096
    assert eb.shape == (3, L + 1, L + 1)
    np.testing.assert_allclose(eb[2], eb[0] - eb[1], atol=1e-12 * np.abs(eb[0]).max())
    assert not eb[:, :2].any() and not eb[:, :, :2].any()      # spin 2 starts at l = 2
    m00 = hx.mixmat(wl, spin=(0, 0))
    s = m00 * (2 * ell + 1)[:, None]            # detailed balance of the coupling
    np.testing.assert_allclose(s, s.T, atol=1e-11 * np.abs(s).max())
    # the oracle's 3j-recursion route on a corner of the matrix (rows < 24)
    ref = oracle.mixmat_eb(wl, l1max=23, l2max=200)
    np.testing.assert_allclose(eb[:, :24, :201], ref, atol=1e-12 * np.abs(ref).max())
    # full-sky mask: identity at this size too
    one = np.zeros(L + 1)
    one[0] = 4 * np.pi
    ident = hx.mixmat(one, spin=(0, 0))
    np.testing.assert_allclose(np.diag(ident), 1.0, atol=1e-11)
    assert np.abs(ident - np.diag(np.diag(ident))).max() <= 1e-11
