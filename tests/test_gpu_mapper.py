"""GPU tests of the drop-in boundary: HipHealpixMapper.transform and the two-point driver,
restating the reference's own tests (tests/test_healpy.py:81-163, tests/test_twopoint.py)."""

import numpy as np
import pytest

from helpers import idx, key_str

pytestmark = pytest.mark.gpu


def test_transform_metadata_and_shapes():
    import heracles_amd as hx
    from heracles_amd import HipHealpixMapper, update_metadata

    rng = np.random.default_rng(50)
    nside = 32
    npix = 12 * nside**2
    mapper = HipHealpixMapper(nside, deconvolve=False)
    nlm = (mapper.lmax + 1) * (mapper.lmax + 2) // 2
    m = rng.standard_normal(npix)
    update_metadata(m, spin=0, nside=nside, a=1)
    alms = mapper.transform(m, spin=0)
    assert alms.shape == (nlm,) and alms.dtype == np.complex128
    assert alms.dtype.metadata["spin"] == 0 and alms.dtype.metadata["a"] == 1
    assert alms.dtype.metadata["nside"] == nside and alms.dtype.metadata["deconv"] is False
    m = rng.standard_normal((2, npix))
    update_metadata(m, spin=2, nside=nside, b=2)
    alms = mapper.transform(m, spin=2)
    assert alms.shape == (2, nlm)
    assert alms.dtype.metadata["spin"] == 2 and alms.dtype.metadata["b"] == 2
    assert isinstance(hx.get_plan(nside, mapper.lmax).scratch_bytes, int)


def test_transform_deconvolve_rule(monkeypatch):
    """With the SHT stubbed to ones, each m-block equals 1/pw[m:], pw[:2] := 1 for spin 2
    (tests/test_healpy.py:119-163)."""
    from heracles_amd import HipHealpixMapper, sht, update_metadata

    nside, lmax = 32, 48
    npix, nlm = 12 * nside**2, (lmax + 1) * (lmax + 2) // 2
    pw0 = 1.0 / (1.0 + 1e-4 * np.arange(lmax + 1) ** 2)
    pw2 = 1.0 / (1.0 + 2e-4 * np.arange(lmax + 1) ** 2)
    real = sht.Plan.map2alm

    def stub(self, maps, spin=0, **kw):
        ones = np.ones(maps.shape)
        out = real(self, 0 * maps, spin, **{**kw, "niter": 0})
        base = np.ones(out.shape, dtype=complex)
        fl = kw.get("fl")
        if fl is not None:
            for m in range(lmax + 1):
                s = idx(lmax, m, m)
                base[..., s : s + lmax - m + 1] *= fl[m:]
        assert ones.shape[-1] == npix and np.abs(out).max() == 0
        return base

    monkeypatch.setattr(sht.Plan, "map2alm", stub)
    mapper = HipHealpixMapper(nside, lmax, deconvolve=True, pixwin=(pw0, pw2))
    data = np.zeros(npix)
    update_metadata(data, spin=0)
    alm = mapper.transform(data, spin=0)
    assert alm.shape == (nlm,)
    stop = 0
    p2 = pw2.copy()
    p2[:2] = 1.0
    for m in range(lmax + 1):
        start, stop = stop, stop + lmax - m + 1
        np.testing.assert_array_equal(alm[start:stop], 1.0 / pw0[m:])
    data = np.zeros((2, npix))
    update_metadata(data, spin=2)
    alm = mapper.transform(data, spin=2)
    assert alm.shape == (2, nlm)
    stop = 0
    for m in range(lmax + 1):
        start, stop = stop, stop + lmax - m + 1
        np.testing.assert_array_equal(alm[0, start:stop], 1.0 / p2[m:])
        np.testing.assert_array_equal(alm[1, start:stop], 1.0 / p2[m:])


def test_maps_to_cls_end_to_end(oracle):
    """maps -> alms -> all spectra on the GPU == the same chain on the oracle."""
    import heracles_amd as hx
    from heracles_amd import HipHealpixMapper, update_metadata

    rng = np.random.default_rng(50)
    nside, lmax = 32, 48
    mapper = HipHealpixMapper(nside, lmax, deconvolve=False, niter=0)
    maps, alms = {}, {}
    for i in (0, 1):
        p = mapper.create(spin=0)
        p[:] = rng.standard_normal(p.shape)
        update_metadata(p, fsky=0.5, musq=1.2, dens=3.4)
        g = mapper.create(2, spin=2)
        g[:] = rng.standard_normal(g.shape)
        maps["POS", i], maps["SHE", i] = p, g
    out = mapper.transform_many(list(maps.values()), [0, 2, 0, 2])
    for k, a in zip(maps, out):
        alms[k] = a
        ref = oracle.map2alm(np.asarray(maps[k]), nside, lmax, spin=a.dtype.metadata["spin"])
        np.testing.assert_allclose(a, ref, atol=1e-11 * np.abs(ref).max())
        single = mapper.transform(maps[k], spin=a.dtype.metadata["spin"])
        np.testing.assert_array_equal(np.asarray(single), np.asarray(a))
    cls = hx.angular_power_spectra(alms)
    assert len(cls) == 10
    for (k1, k2, i1, i2), res in cls.items():
        ref = oracle.alm2cl(np.asarray(alms[k1, i1]), np.asarray(alms[k2, i2]))
        md = res.array.dtype.metadata
        if "bias" in md:
            lmin = max(md["spin_1"], md["spin_2"])
            if md["spin_1"] == 2:
                ref[0, 0, lmin:] -= md["bias"]
                ref[1, 1, lmin:] -= md["bias"]
            else:
                ref[..., lmin:] -= md["bias"]
        np.testing.assert_allclose(res.array, ref, rtol=1e-10, atol=1e-14)


def test_angular_power_spectra_golden(golden):
    import heracles_amd as hx

    alms = {}
    for n, i in [("POS", 0), ("POS", 1), ("SHE", 0), ("SHE", 1)]:
        a = np.array(golden[f"alm/{key_str((n, i))}"])
        md = {"nside": 32, "spin": 0 if n == "POS" else 2, "geometry": "plain", "kernel": "plain"}
        if i == 0:
            md.update(fsky=0.5, musq=1.2, dens=3.4)
        a.dtype = np.dtype(a.dtype, metadata=md)
        alms[n, i] = a
    cls = hx.angular_power_spectra(alms)
    assert [key_str(k) for k in cls] == list(golden["aps/plain/keys"])
    for k, v in cls.items():
        np.testing.assert_allclose(np.asarray(v.array), golden[f"aps/plain/cl/{key_str(k)}"], rtol=1e-12, atol=1e-14)
