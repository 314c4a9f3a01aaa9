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
    # heracles.transform's interface over the same maps (heracles_amd.transform: one batched call per mapper)
    from types import SimpleNamespace

    fields = {"POS": SimpleNamespace(spin=0, mapper_or_error=mapper), "SHE": SimpleNamespace(spin=2, mapper_or_error=mapper)}
    again = hx.transform(fields, maps)
    assert list(again) == list(maps)
    for k in maps:
        np.testing.assert_array_equal(np.asarray(again[k]), np.asarray(alms[k]))
        assert again[k].dtype.metadata["spin"] == alms[k].dtype.metadata["spin"]
    # ... with the alms kept in HBM (DeviceArrays): the same spectra without a PCIe round trip of the alms
    resident = hx.transform(fields, maps, device="cuda")
    assert all(isinstance(v, hx.DeviceArray) for v in resident.values())
    cls_res = hx.angular_power_spectra(resident)
    cls = hx.angular_power_spectra(alms)
    assert len(cls) == 10 and list(cls_res) == list(cls)
    for k in cls:
        np.testing.assert_array_equal(np.asarray(cls_res[k].array), np.asarray(cls[k].array))
        assert cls_res[k].array.dtype.metadata == cls[k].array.dtype.metadata
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


# ---- catalogue -> map accumulation (hx_ang2pix_ring / hx_map_values) ---------------------
@pytest.mark.parametrize("nside", [1, 2, 3, 16, 64, 4096])
def test_ang2pix_matches_oracle(oracle, nside):
    from heracles_amd.mapper import ang2pix_ring

    rng = np.random.default_rng(nside)
    n = 200_000
    lon = rng.uniform(-360, 720, n)
    lat = np.degrees(np.arcsin(rng.uniform(-1, 1, n)))
    lon[:6] = [0.0, 90.0, 359.999, 360.0, -0.0, 720.0]
    lat[:6] = [90.0, 90.0, -90.0, 0.0, 0.0, 41.8103148957786]
    lat[6:2000] = rng.uniform(89.0, 90.0, 1994) * rng.choice([-1, 1], 1994)  # sin(theta) branch near the poles
    got = ang2pix_ring(nside, lon, lat)
    assert got.dtype == np.int64
    np.testing.assert_array_equal(got, oracle.ang2pix_ring(nside, lon, lat))
    if nside <= 64:
        theta, phi = oracle.pix2ang(nside)
        np.testing.assert_array_equal(ang2pix_ring(nside, np.degrees(phi), 90.0 - np.degrees(theta)),
                                      np.arange(12 * nside**2))


def test_invalid_points_raise_before_any_scatter():
    """healpy.ang2pix raises ValueError for |lat| > 90 and for NaN (check_theta_valid), so the reference never
    scatters such a page; here the device flags them and nothing is written."""
    import heracles_amd as hx
    from heracles_amd.mapper import ang2pix_ring, map_values

    nside = 16
    lon = np.array([10.0, 20.0, 30.0, 40.0])
    for bad in (90.05, -91.0, np.nan, 1e300):
        lat = np.array([0.0, bad, 45.0, -45.0])
        with pytest.raises(ValueError):
            ang2pix_ring(nside, lon, lat)
        maps = np.full((2, 12 * nside**2), 7.0)
        with pytest.raises(ValueError):
            map_values(nside, lon, lat, maps, np.ones((2, 4)))
        assert (maps == 7.0).all()
    with pytest.raises(ValueError):
        ang2pix_ring(nside, np.array([np.inf, 0.0, 0.0, 0.0]), np.zeros(4))
    # the poles and the seam are valid
    ang2pix_ring(nside, np.array([0.0, 360.0, -720.0, 1e6]), np.array([90.0, -90.0, 0.0, 12.0]))
    m = hx.HipHealpixMapper(nside)
    d = m.create()
    with pytest.raises(ValueError):
        m.map_values(lon, np.array([0.0, 0.0, 95.0, 0.0]), d, np.ones(4))
    assert not d.any()


def test_map_values_reference_case():
    """tests/test_healpy.py:44-77 with the GPU mapper: order-exact sums."""
    from heracles_amd import HipHealpixMapper
    from heracles_amd.mapper import ang2pix_ring

    nside = 32
    npix = 12 * nside**2
    rng = np.random.default_rng(50)
    mapper = HipHealpixMapper(nside)
    size = 1000
    lon = rng.uniform(0, 360, size=size)
    lat = np.degrees(np.arcsin(rng.uniform(-1, 1, size=size)))
    x, y = rng.standard_normal(size), rng.standard_normal(size)
    ipix = ang2pix_ring(nside, lon, lat)
    m = mapper.create()
    mapper.map_values(lon, lat, m, x)
    expected = np.zeros(npix)
    np.add.at(expected, ipix, x)
    np.testing.assert_array_equal(m, expected)
    m = mapper.create(2)
    mapper.map_values(lon, lat, m, np.stack([x, y]))
    expected = np.zeros((2, npix))
    np.add.at(expected[0], ipix, x)
    np.add.at(expected[1], ipix, y)
    np.testing.assert_array_equal(m, expected)
    # accumulates on top of what is there; big-endian inputs are byteswapped (heracles/healpy.py:43-55)
    mapper.map_values(lon.astype(">f8"), lat.astype(">f8"), m, np.stack([x, y]).astype(">f8"))
    np.add.at(expected[0], ipix, x)
    np.add.at(expected[1], ipix, y)
    np.testing.assert_array_equal(m, expected)
    mapper.map_values(lon[:0], lat[:0], m, np.stack([x, y])[:, :0])  # empty page
    np.testing.assert_array_equal(m, expected)


def test_map_values_crowded_pixels_ordered_and_atomic(oracle):
    """Many points per pixel, values of very different magnitude: the ordered path is
    bit-identical to the sequential loop, the atomic path only to rounding."""
    import torch
    from heracles_amd.mapper import map_values

    nside = 8
    npix = 12 * nside**2
    rng = np.random.default_rng(3)
    n = 300_000
    lon = rng.uniform(0, 360, n)
    lat = np.degrees(np.arcsin(rng.uniform(-1, 1, n)))
    lon[: n // 3] = 12.5
    lat[: n // 3] = -33.0  # a third of the catalogue in one pixel
    vals = rng.standard_normal((3, n)) * 10.0 ** rng.integers(-8, 8, (3, n))
    exp = rng.standard_normal((3, npix))
    got = exp.copy()
    oracle.map_values(nside, lon, lat, exp, vals)
    map_values(nside, lon, lat, got, vals)
    np.testing.assert_array_equal(got, exp)
    dev = torch.zeros((3, npix), dtype=torch.float64, device="cuda")
    map_values(nside, torch.as_tensor(lon).cuda(), torch.as_tensor(lat).cuda(), dev, torch.as_tensor(vals).cuda())
    ref = np.zeros((3, npix))
    oracle.map_values(nside, lon, lat, ref, vals)
    np.testing.assert_array_equal(dev.cpu().numpy(), ref)
    dev.zero_()
    map_values(nside, lon, lat, dev, vals, ordered=False)
    scale = np.zeros((3, npix))
    oracle.map_values(nside, lon, lat, scale, np.abs(vals))
    assert np.all(np.abs(dev.cpu().numpy() - ref) <= 1e-12 * scale + 1e-300)


def test_map_values_full_size_checksum():
    """nside=4096: sum over pixels == sum over the catalogue (exact for integer-valued
    weights), hit counts == np.bincount of the indices."""
    import torch
    from heracles_amd.mapper import ang2pix_ring, map_values

    nside = 4096
    npix = 12 * nside**2
    g = torch.Generator(device="cuda").manual_seed(11)
    n = 20_000_000
    lon = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 360.0
    lat = torch.rad2deg(torch.asin(torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1))
    w = torch.randint(-1000, 1000, (2, n), device="cuda", generator=g).to(torch.float64)
    w[0] = 1.0
    maps = torch.zeros((2, npix), dtype=torch.float64, device="cuda")
    map_values(nside, lon, lat, maps, w)
    assert float(maps[0].sum()) == n and float(maps[1].sum()) == float(w[1].sum())
    ipix = ang2pix_ring(nside, lon, lat)
    assert int(ipix.min()) >= 0 and int(ipix.max()) < npix
    assert torch.equal(maps[0], torch.bincount(ipix, minlength=npix).to(torch.float64))


# ---- resample / ud_grade (hx_ud_grade) ---------------------------------------------------
@pytest.mark.parametrize("nside_in,nside_out", [(16, 8), (16, 4), (32, 2), (64, 1), (8, 8), (4, 16), (1, 8), (128, 16)])
def test_ud_grade_matches_oracle(oracle, nside_in, nside_out):
    """Bit-identical to healpy's arithmetic (numpy pairwise sums over the NEST children)."""
    from heracles_amd.mapper import ud_grade

    rng = np.random.default_rng(nside_in * 100 + nside_out)
    m = rng.standard_normal((3, 12 * nside_in**2)) * 10.0 ** rng.integers(-5, 5, (3, 12 * nside_in**2))
    m[1, rng.integers(0, m.shape[1], m.shape[1] // 3)] = oracle.UNSEEN   # masked pixels
    m[2, :7] = [np.nan, np.inf, -np.inf, oracle.UNSEEN, 0.0, -0.0, 1e300]
    got = ud_grade(m, nside_out)
    exp = oracle.ud_grade(m, nside_out)
    assert got.shape == exp.shape == (3, 12 * nside_out**2)
    np.testing.assert_array_equal(got, exp)
    np.testing.assert_array_equal(ud_grade(m[0], nside_out), exp[0])


def test_mapper_resample_and_errors(oracle):
    import torch
    from heracles_amd import HipHealpixMapper
    from heracles_amd.mapper import ud_grade

    rng = np.random.default_rng(5)
    m = rng.standard_normal(12 * 64**2)
    mapper = HipHealpixMapper(16, dtype=np.float32)
    out = mapper.resample(m)
    assert out.dtype == np.float32 and out.shape == (12 * 16**2,)
    np.testing.assert_array_equal(out, oracle.ud_grade(m, 16).astype(np.float32))
    dev = ud_grade(torch.as_tensor(m).cuda(), 256)
    assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), oracle.ud_grade(m, 256))
    with pytest.raises(ValueError, match="power of 2"):
        ud_grade(m, 12)
    with pytest.raises(ValueError, match="12\\*nside"):
        ud_grade(m[:-1], 16)


def test_ud_grade_full_size_properties():
    """nside 4096 -> 1024 -> 4096 on the device: the mean is conserved, replication is exact."""
    import torch
    from heracles_amd.mapper import ud_grade

    g = torch.Generator(device="cuda").manual_seed(3)
    m = torch.randint(-2**20, 2**20, (12 * 4096**2,), device="cuda", generator=g).to(torch.float64)
    d = ud_grade(m, 1024)
    assert float(d.sum()) * 16 == float(m.sum())        # integer-valued pixels: every sum is exact
    u = ud_grade(d, 4096)
    assert torch.equal(ud_grade(u, 1024), d)


# ---- DiscreteMapper.resample (hx_alm_resample) -------------------------------------------
def test_discrete_resample_reference_case():
    """tests/test_ducc.py:12-45 with the HIP mapper: identity, truncation, zero-padding."""
    import torch
    from heracles_amd import HipDiscreteMapper, alm_resample

    lmax = 200
    alm = np.concatenate([np.arange(m, lmax + 1) for m in range(lmax + 1)], dtype=complex)
    np.testing.assert_array_equal(HipDiscreteMapper(lmax).resample(alm), alm)
    lmax_out = lmax // 2
    out = HipDiscreteMapper(lmax_out).resample(alm)
    assert out.shape == ((lmax_out + 1) * (lmax_out + 2) // 2,)
    i = j = 0
    for m in range(lmax_out + 1):
        i, j = j, j + lmax_out - m + 1
        np.testing.assert_array_equal(out[i:j], np.arange(m, lmax_out + 1))
    lmax_out = lmax * 2
    out = HipDiscreteMapper(lmax_out).resample(alm)
    assert out.shape == ((lmax_out + 1) * (lmax_out + 2) // 2,)
    i = j = 0
    for m in range(lmax + 1):
        i, j = j, j + lmax_out - m + 1
        np.testing.assert_array_equal(out[i:j], np.pad(np.arange(m, lmax + 1), (0, lmax_out - lmax)))
    np.testing.assert_array_equal(out[j:], 0.0)
    # leading dimensions, dtype, device tensors
    rng = np.random.default_rng(2)
    a = rng.standard_normal((2, 3, 21 * 22 // 2)) + 1j * rng.standard_normal((2, 3, 21 * 22 // 2))
    o = HipDiscreteMapper(9, dtype=np.complex64).resample(a)
    assert o.shape == (2, 3, 55) and o.dtype == np.complex64
    d = alm_resample(torch.as_tensor(a).cuda(), 30)
    back = alm_resample(d, 20)
    assert d.is_cuda and np.array_equal(back.cpu().numpy(), a)
    with pytest.raises(ValueError):
        alm_resample(a[..., :-1], 5)


def test_device_resident_catalogue_to_alm(oracle):
    """catalogue page -> map_values on the device -> transform on the device: nothing crosses PCIe
    until the alms are read; same numbers as the host path and the oracle."""
    import torch
    from heracles_amd import HipHealpixMapper

    nside, lmax = 32, 48
    rng = np.random.default_rng(21)
    n = 50_000
    lon = rng.uniform(0, 360, n)
    lat = np.degrees(np.arcsin(rng.uniform(-1, 1, n)))
    w = rng.standard_normal((2, n))
    mapper = HipHealpixMapper(nside, lmax, deconvolve=False, niter=0)
    dmaps = torch.zeros((2, 12 * nside**2), dtype=torch.float64, device="cuda")
    for sl in (slice(0, n // 2), slice(n // 2, n)):          # two pages
        mapper.map_values(lon[sl], lat[sl], dmaps, w[:, sl])
    hmaps = mapper.create(2, spin=2)
    mapper.map_values(lon, lat, hmaps, w)
    np.testing.assert_array_equal(dmaps.cpu().numpy(), hmaps)
    dalm = mapper.transform(dmaps, spin=2)
    assert dalm.is_cuda and dalm.shape == (2, (lmax + 1) * (lmax + 2) // 2)
    halm = mapper.transform(hmaps, spin=2)
    np.testing.assert_array_equal(dalm.cpu().numpy(), np.asarray(halm))
    ref = oracle.map2alm(np.asarray(hmaps), nside, lmax, spin=2)
    assert np.abs(np.asarray(halm) - ref).max() <= 1e-11 * np.abs(ref).max()
    with pytest.raises(ValueError, match="spin-2 maps"):
        mapper.transform(dmaps[0], spin=2)


def test_hx_copy_between_host_and_device():
    """hx_copy: host -> device -> device -> host through the library's staging pipeline, sizes below and above its chunk."""
    import torch

    import heracles_amd as hx

    rng = np.random.default_rng(1)
    for n in (1000, 5_000_000):
        src = rng.standard_normal(n)
        d1 = torch.empty(n, dtype=torch.float64, device="cuda")
        d2 = torch.empty(n, dtype=torch.float64, device="cuda")
        back = np.empty(n)
        hx._lib.copy(d1, src)
        hx._lib.copy(d2, d1)
        hx._lib.copy(back, d2)
        np.testing.assert_array_equal(back, src)
    with pytest.raises(ValueError):
        hx._lib.copy(np.empty(3), np.empty(4))


def test_transform_many_takes_device_maps_and_warns_like_transform(oracle):
    """``transform_many`` with device-resident maps and the default ``device=None`` returns device tensors, as
    ``transform`` does for such maps; ``heracles_amd.transform`` accepts them (no dtype metadata to read); the unit-weight note
    of ``transform`` is not lost on the batched route."""
    import types
    import warnings

    import torch

    import heracles_amd as hx
    from heracles_amd import mapper as hm

    nside, lmax = 16, 24
    rng = np.random.default_rng(5)
    t = rng.standard_normal(12 * nside**2)
    qu = rng.standard_normal((2, 12 * nside**2))
    mp = hx.HipHealpixMapper(nside, lmax, deconvolve=False, niter=0)
    hm._warned_unit_weights = False
    with pytest.warns(RuntimeWarning, match="unit quadrature"):
        got = mp.transform_many([torch.as_tensor(t).cuda(), torch.as_tensor(qu).cuda()], [0, 2])
    assert all(torch.is_tensor(a) and a.is_cuda for a in got)
    r0 = oracle.map2alm(t[None], nside, lmax, spin=0)[0]
    r2 = oracle.map2alm(qu, nside, lmax, spin=2)
    np.testing.assert_allclose(got[0].cpu().numpy(), r0, atol=1e-11 * np.abs(r0).max())
    np.testing.assert_allclose(got[1].cpu().numpy(), r2, atol=1e-11 * np.abs(r2).max())
    fields = {"POS": types.SimpleNamespace(spin=0, mapper_or_error=mp)}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = hx.transform(fields, {("POS", 0): torch.as_tensor(t).cuda()})
    np.testing.assert_allclose(out["POS", 0].cpu().numpy(), r0, atol=1e-11 * np.abs(r0).max())


def test_map_values_wide_keys_path_in_a_child_process(tmp_path):
    """Pixel indices beyond 32 bits (nside > 16384) keep 64-bit keys through every pass of the sort (hx_sort.h: radix_sort_pairs<long long>);
    HX_SORT_WIDE=1 sends an ordinary catalogue down that path: order-exact like the narrowed one (the switch is read once per process)."""
    import os
    import subprocess
    import sys

    code = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from oracle import hxoracle as ho
from heracles_amd.mapper import map_values
nside = 16
rng = np.random.default_rng(8)
n = 70001
lon = rng.uniform(0, 360, n); lat = np.degrees(np.arcsin(rng.uniform(-1, 1, n)))
lon[:9000] = 77.0; lat[:9000] = 12.0
vals = rng.standard_normal((2, n)) * 10.0 ** rng.integers(-6, 6, (2, n))
exp = np.zeros((2, 12 * nside**2)); got = exp.copy()
ho.map_values(nside, lon, lat, exp, vals)
map_values(nside, lon, lat, got, vals)
assert np.array_equal(got, exp)
print("wide ok")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HX_SORT_WIDE="1"), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "wide ok" in res.stdout, res.stderr[-1500:]
