"""Host-side logic of heracles_amd that needs no GPU: key handling, metadata, bias rule,
mask -> field fan-out.  Arithmetic kernels are replaced by the oracle (as a checker stub)
so the Python drivers can be compared with the reference's golden outputs on CPU."""

import os
import types

import numpy as np
import pytest

import heracles_amd
from heracles_amd import twopoint as tp
from helpers import key_str

NAMES = [("POS", 0), ("POS", 1), ("SHE", 0), ("SHE", 1)]


@pytest.fixture
def cpu_kernels(monkeypatch, oracle):
    """Stub the HIP entry points with the oracle so driver logic runs without a GPU."""

    def pairs(comps, plist, lmax_out):
        out = np.zeros((len(plist), lmax_out + 1))
        for n, (i, j) in enumerate(plist):
            out[n] = oracle.alm2cl(np.asarray(comps[i]), np.asarray(comps[j]), lmax=lmax_out)
        return out

    monkeypatch.setattr(tp, "alm2cl_pairs", pairs)
    return oracle


def golden_alms(golden, with_bias=True):
    alms = {}
    for n, i in NAMES:
        a = np.array(golden[f"alm/{key_str((n, i))}"])
        md = {"nside": 32, "spin": 0 if n == "POS" else 2}
        if with_bias and i == 0:
            md.update(fsky=0.5, musq=1.2, dens=3.4)
        md.update(geometry="plain", kernel="plain")
        a.dtype = np.dtype(a.dtype, metadata=md)
        alms[n, i] = a
    return alms


def test_alm2lmax(golden):
    for n, v in zip(golden["alm2lmax_sizes"], golden["alm2lmax_values"]):
        assert tp.alm2lmax(np.zeros(int(n))) == v


def test_alm2cl_driver_shapes(cpu_kernels, golden):
    from itertools import combinations_with_replacement

    for k1, k2 in combinations_with_replacement(NAMES, 2):
        a, b = golden[f"alm/{key_str(k1)}"], golden[f"alm/{key_str(k2)}"]
        ref = golden[f"alm2cl/{key_str(k1)}/{key_str(k2)}"]
        out = tp.alm2cl(a, b)
        assert out.shape == ref.shape
        np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-14)
    out = tp.alm2cl(golden["uneq/a1"], golden["uneq/a2"], lmax=20)
    np.testing.assert_allclose(out, golden["uneq/cl_lmax20"], rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("tag,kw", [
    ("plain", {}), ("nodebias", {"debias": False}), ("lmax16", {"lmax": 16}),
    ("incl", {"include": [("POS", "SHE", ..., ...)]}), ("excl", {"exclude": [("SHE", "SHE")]}),
])
def test_angular_power_spectra_vs_reference(cpu_kernels, golden, tag, kw):
    cls = tp.angular_power_spectra(golden_alms(golden), **kw)
    assert [key_str(k) for k in cls] == list(golden[f"aps/{tag}/keys"])  # order included
    for k, v in cls.items():
        np.testing.assert_allclose(np.asarray(v.array), golden[f"aps/{tag}/cl/{key_str(k)}"], rtol=1e-12, atol=1e-14)
        md = v.array.dtype.metadata or {}
        assert sorted(f"{a}={md[a]!r}" for a in md) == list(golden[f"aps/{tag}/md/{key_str(k)}"])
        assert v.axis == (v.ndim - 1,)
        assert v.spin == (md["spin_1"], md["spin_2"])


def test_angular_power_spectra_cross_and_reversed(cpu_kernels, golden):
    alms = golden_alms(golden)
    a1 = {k: v for k, v in alms.items() if k[1] == 0}
    a2 = {k: v for k, v in alms.items() if k[1] == 1}
    cls = tp.angular_power_spectra(a1, a2)
    assert [key_str(k) for k in cls] == list(golden["aps/cross/keys"])
    for k, v in cls.items():
        np.testing.assert_allclose(np.asarray(v.array), golden[f"aps/cross/cl/{key_str(k)}"], rtol=1e-12, atol=1e-14)
    rev = dict(reversed(list(alms.items())))
    cls = tp.angular_power_spectra(rev)
    assert [key_str(k) for k in cls] == list(golden["aps/rev/keys"])
    for k, v in cls.items():
        np.testing.assert_allclose(np.asarray(v.array), golden[f"aps/rev/cl/{key_str(k)}"], rtol=1e-12, atol=1e-14)


def test_angular_power_spectra_bias_rule(cpu_kernels):
    # restates tests/test_twopoint.py:154-193 of the reference
    lmax = 32
    size = (lmax + 1) * (lmax + 2) // 2
    fsky, musq, dens = 0.5, 1.2, 3.4
    a = np.zeros(size, dtype=complex)
    a.dtype = np.dtype(a.dtype, metadata={"spin": 0, "fsky": fsky, "musq": musq, "dens": dens})
    cls = tp.angular_power_spectra({("F", 0): a}, debias=False)
    assert cls["F", "F", 0, 0].dtype.metadata["bias"] == pytest.approx(fsky * musq / dens)
    b = np.zeros((2, size), dtype=complex)
    b.dtype = np.dtype(b.dtype, metadata={"spin": 2, "fsky": fsky, "musq": musq, "dens": dens})
    cls2 = tp.angular_power_spectra({("G", 0): b}, debias=False)
    assert cls2["G", "G", 0, 0].dtype.metadata["bias"] == pytest.approx(0.5 * fsky * musq / dens)
    cross = tp.angular_power_spectra({("F", 0): a, ("F", 1): a.copy()}, debias=False)
    assert "bias" not in (cross["F", "F", 0, 1].dtype.metadata or {})
    c = np.zeros(size, dtype=complex)
    c.dtype = np.dtype(c.dtype, metadata={"spin": 0})
    ext = tp.angular_power_spectra({("F", 0): c}, debias=False)
    assert "bias" not in (ext["F", "F", 0, 0].dtype.metadata or {})
    d = np.zeros(size, dtype=complex)
    with pytest.raises(ValueError, match="missing spin metadata"):
        tp.angular_power_spectra({("F", 0): d})


def test_debias_cl_vs_reference(golden):
    cases = {"a": (1.23, {}), "c": (None, {"bias": 4.56, "spin_2": 2}),
             "d": (7.89, {"spin_1": 2, "spin_2": 2}), "e": (7.89, {"spin_1": 0, "spin_2": 0})}
    for k, (bias, md) in cases.items():
        arr = np.array(golden[f"debias/{k}/in"])
        arr.dtype = np.dtype(arr.dtype, metadata=md)
        out = tp._debias_cl(arr, bias)
        np.testing.assert_array_equal(out, golden[f"debias/{k}/out"])
        assert out is not arr


def test_debias_healpix_pixwin_rule():
    # restates tests/test_twopoint.py:243-290 with an explicit pixel window table
    lmax = 99
    pw0 = 1.0 / (1.0 + 1e-4 * np.arange(lmax + 1) ** 2)
    pw2 = 1.0 / (1.0 + 2e-4 * np.arange(lmax + 1) ** 2)
    md1 = {"kernel_1": "healpix", "nside_1": 64, "kernel_2": "healpix", "nside_2": 64, "spin_2": 2}
    md2 = {"kernel_1": "healpix", "nside_1": 64, "spin_2": 2}
    md3 = dict(md1, deconv_2=False)
    cls = {i: np.zeros((2, 100), dtype=np.dtype(float, metadata=md)) for i, md in ((1, md1), (2, md2), (3, md3))}
    tp.debias_cls(cls, {1: 1.23, 2: 4.56, 3: 7.89}, inplace=True, pixwin=(pw0, pw2))
    np.testing.assert_array_equal(cls[1][:, :2], 0.0)
    np.testing.assert_array_equal(cls[1][0, 2:], -1.23 / pw0[2:] / pw2[2:])
    np.testing.assert_array_equal(cls[2][1, 2:], -4.56 / pw0[2:])
    np.testing.assert_array_equal(cls[3][0, 2:], -7.89 / pw0[2:])


def test_mixing_matrices_driver_vs_reference(golden):
    """Keys, order, spin dispatch and arguments of the reference's loop (golden: the call list its mocked convolvecl
    functions saw, tests/test_twopoint.py:316-383) -- here the requests reach ONE context per (l1max, l2max, l3max)."""
    calls = []

    def recorder(cl, l1max, l2max, l3max, spin):
        name = "mixmat" if 0 in spin else "mixmat_eb"
        calls.append((name, tuple(spin), l1max, l2max, l3max))
        n = len(cl)
        return np.zeros((n, n)) if name == "mixmat" else np.zeros((3, n, n))

    cl = np.arange(21.0)
    mm_cls = {("VIS", "VIS", 0, 1): cl, ("VIS", "WHT", 0, 1): cl, ("WHT", "VIS", 0, 1): cl,
              ("WHT", "WHT", 0, 1): cl, ("X", "Y", 0, 1): cl, ("WHT", "WHT", 1, 1): cl}
    flds = {"POS": types.SimpleNamespace(mask="VIS", spin=0), "SHE": types.SimpleNamespace(mask="WHT", spin=2),
            "POS2": types.SimpleNamespace(mask="VIS", spin=0), "NOMASK": types.SimpleNamespace(mask=None, spin=0)}
    mms = tp.mixing_matrices(flds, mm_cls, l1max=10, l2max=12, l3max=20, context=recorder)
    assert [key_str(k) for k in mms] == list(golden["mm/keys"])
    assert [repr(c) for c in calls] == list(golden["mm/calls"])
    assert [repr(v.axis) for v in mms.values()] == list(golden["mm/axis"])
    assert [key_str(t) for t, _, _ in tp.mixing_requests(flds, mm_cls)] == list(golden["mm/keys"])


def test_core_helpers():
    from heracles_amd.core import Result, TocDict, toc_match, update_metadata

    a = np.zeros(4)
    update_metadata(a, x=1)
    update_metadata(a, y=2)
    assert a.dtype.metadata == {"x": 1, "y": 2}
    assert toc_match(("a", "b", 0, 1), include=[("a", ..., 0)])
    assert not toc_match(("a", "b", 0, 1), exclude=[(..., "b")])
    d = TocDict({("a", "b", 0, 1): 1, ("a", "c", 0, 1): 2, ("x", "b", 1, 1): 3})
    assert d["a"] == {("a", "b", 0, 1): 1, ("a", "c", 0, 1): 2}
    assert d[..., "b"] == {("a", "b", 0, 1): 1, ("x", "b", 1, 1): 3}
    with pytest.raises(KeyError):
        d["zzz"]
    r = Result(np.zeros((2, 5)), spin=(0, 2), axis=-1)
    assert r.axis == (1,) and r.shape == (2, 5)
    assert Result(np.zeros((3, 4, 5)), axis=-2).axis == (1,)


def test_mapper_interface_and_metadata():
    from heracles_amd import HipHealpixMapper

    nside, lmax = 8, 12
    mapper = HipHealpixMapper(nside, lmax, deconvolve=False)
    assert mapper.nside == nside and mapper.lmax == lmax and mapper.deconvolve is False
    assert HipHealpixMapper(16).lmax == 24 and HipHealpixMapper(16).deconvolve is True
    assert mapper.area == pytest.approx(4 * np.pi / (12 * nside**2))
    for attr in ("area", "create", "map_values", "transform", "resample"):
        assert hasattr(mapper, attr)
    m = mapper.create(1, 2, 3, spin=-3)
    assert m.shape == (1, 2, 3, 12 * nside**2)
    assert m.dtype.metadata == {"geometry": "healpix", "kernel": "healpix", "nside": nside,
                                "lmax": lmax, "deconv": False, "spin": -3}
    with pytest.raises(NotImplementedError, match="spin-1 maps not yet supported"):
        mapper.transform(mapper.create(), spin=1)


def test_oracle_ang2pix_and_map_values(oracle):
    """The numpy restatement of ang2pix / map_values that the GPU path is checked against:
    pix2ang round trip of every pixel centre (any NSIDE, not only powers of two), range,
    locality, the pole / wrap-around cases and the reference's own map_values expectation
    (tests/test_healpy.py:57-77)."""
    for nside in (1, 2, 3, 4, 5, 16, 32):
        theta, phi = oracle.pix2ang(nside)
        ipix = oracle.ang2pix_ring(nside, np.degrees(phi), 90.0 - np.degrees(theta))
        np.testing.assert_array_equal(ipix, np.arange(12 * nside**2))
    rng = np.random.default_rng(7)
    nside = 16
    lon = rng.uniform(-360, 720, 5000)
    lat = np.degrees(np.arcsin(rng.uniform(-1, 1, 5000)))
    ipix = oracle.ang2pix_ring(nside, lon, lat)
    assert ipix.min() >= 0 and ipix.max() < 12 * nside**2
    np.testing.assert_array_equal(ipix, oracle.ang2pix_ring(nside, np.mod(lon, 360.0), lat))
    theta, phi = oracle.pix2ang(nside)
    vec = np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], 1)
    t, p = np.radians(90 - lat), np.radians(lon)
    v = np.stack([np.sin(t) * np.cos(p), np.sin(t) * np.sin(p), np.cos(t)], 1)
    d_own = np.arccos(np.clip((v * vec[ipix]).sum(1), -1, 1))
    assert d_own.max() < 1.5 * np.sqrt(4 * np.pi / (12 * nside**2))
    # poles and the lon = 0 / 360 seam
    assert oracle.ang2pix_ring(nside, [0.0, 90.0, 359.999], [90.0, 90.0, 90.0]).tolist() == [0, 1, 3]
    npix = 12 * nside**2
    assert oracle.ang2pix_ring(nside, [0.0, 359.999], [-90.0, -90.0]).tolist() == [npix - 4, npix - 1]
    assert oracle.ang2pix_ring(nside, [360.0], [0.0])[0] == oracle.ang2pix_ring(nside, [0.0], [0.0])[0]
    x, y = rng.standard_normal(5000), rng.standard_normal(5000)
    m = np.zeros((2, npix))
    oracle.map_values(nside, lon, lat, m, np.stack([x, y]))
    exp = np.zeros((2, npix))
    for j, i in enumerate(ipix):  # the loop of heracles/healpy.py:58-66
        exp[0, i] += x[j]
        exp[1, i] += y[j]
    np.testing.assert_array_equal(m, exp)


def test_oracle_ring_nest_and_ud_grade(oracle):
    """RING<->NEST restatement: bijection, consistency with ang2pix across resolutions (a NEST
    parent is index >> 2), and ud_grade against its definition."""
    for ns in (1, 2, 4, 8, 32):
        p = np.arange(12 * ns * ns)
        q = oracle.ring2nest(ns, p)
        assert np.array_equal(np.sort(q), p)
        assert np.array_equal(oracle.nest2ring(ns, q), p)
    rng = np.random.default_rng(11)
    lon = rng.uniform(0, 360, 50000)
    lat = np.degrees(np.arcsin(rng.uniform(-1, 1, 50000)))
    for ns in (2, 16, 256):
        fine, coarse = oracle.ang2pix_ring(ns, lon, lat), oracle.ang2pix_ring(ns // 2, lon, lat)
        assert np.array_equal(oracle.nest2ring(ns // 2, oracle.ring2nest(ns, fine) >> 2), coarse)
    m = rng.standard_normal(12 * 16 * 16)
    d = oracle.ud_grade(m, 8)
    kids = oracle.nest2ring(16, 4 * oracle.ring2nest(8, np.arange(12 * 64))[:, None] + np.arange(4))
    np.testing.assert_allclose(d, m[kids].mean(axis=1), rtol=1e-15, atol=1e-16)
    u = oracle.ud_grade(m, 64)
    assert np.array_equal(oracle.ud_grade(u, 16), m)           # replicate, then average equal values
    assert np.array_equal(oracle.ud_grade(m, 16), m)
    m2 = m.copy()
    m2[kids[5]] = oracle.UNSEEN                                  # a fully masked parent
    m2[kids[7, :2]] = oracle.UNSEEN                              # a half masked parent
    d2 = oracle.ud_grade(m2, 8)
    assert d2[5] == oracle.UNSEEN
    assert d2[7] == (m[kids[7, 2]] + m[kids[7, 3]]) / 2          # masked children enter the sum as zeros
    with pytest.raises(ValueError):
        oracle.ud_grade(m, 12)


def test_discrete_mapper_surface():
    """heracles/ducc.py:40-162 surface: protocol members, metadata of create, identity transform,
    no silent fallback for map_values."""
    from heracles_amd import HipDiscreteMapper

    mapper = HipDiscreteMapper(12, nthreads=4)
    for attr in ("area", "create", "map_values", "transform", "resample"):
        assert hasattr(mapper, attr)
    assert mapper.lmax == 12 and mapper.area == 1.0
    m = mapper.create(2, 3, spin=2)
    assert m.shape == (2, 3, 13 * 14 // 2) and m.dtype == np.complex128 and not m.any()
    assert m.dtype.metadata == {"geometry": "discrete", "kernel": "none", "lmax": 12, "spin": 2}
    assert mapper.transform(m, spin=2) is m
    assert HipDiscreteMapper(4, dtype=np.complex64).create().dtype == np.complex64
    import heracles_amd as hx

    if hx.device_count() == 0:  # no device: the HIP path fails loudly, nothing is computed on the host
        with pytest.raises(hx.HxError):
            mapper.map_values(np.zeros(3), np.zeros(3), mapper.create(), np.zeros(3))


def test_transform_driver_mirrors_heracles_transform():
    """heracles.transform (heracles/mapping.py:113-175) restated around a batched mapper call: keys in the order of the data, the
    spin rule (a map without spin metadata takes its field's, a mismatch is a ValueError), the unknown-field error, maps wrapped in
    objects with ``.array``, ``out=`` filled in place, one ``transform_many`` call per mapper and the per-map fallback."""
    from types import SimpleNamespace

    from heracles_amd import TocDict, transform, update_metadata

    calls = []

    class Batched:
        def transform_many(self, maps, spins):
            calls.append(("many", list(spins)))
            return [np.full(3, 10.0 * s + float(np.asarray(m).ravel()[0])) for m, s in zip(maps, spins)]

    class PerMap:
        def transform(self, m, spin=0):
            calls.append(("one", spin))
            return np.full(2, -1.0 - spin)

    a, b = Batched(), PerMap()
    fields = {"POS": SimpleNamespace(spin=0, mapper_or_error=a), "SHE": SimpleNamespace(spin=2, mapper_or_error=a),
              "VIS": SimpleNamespace(spin=0, mapper=b, mapper_or_error=None)}
    m0, m1, m2, m3 = np.zeros(4) + 1, np.zeros((2, 4)) + 2, np.zeros(4) + 3, np.zeros(4) + 4
    update_metadata(m1, spin=2)
    data = {("SHE", 1): m1, ("POS", 0): m0, ("VIS", 0): SimpleNamespace(array=m3), ("POS", 1): m2}
    out = TocDict()
    res = transform(fields, data, out=out)
    assert res is out and list(out) == list(data)
    assert m0.dtype.metadata["spin"] == 0 and m3.dtype.metadata["spin"] == 0
    assert calls == [("many", [2, 0, 0]), ("one", 0)]
    assert out["SHE", 1][0] == 22.0 and out["POS", 0][0] == 1.0 and out["POS", 1][0] == 3.0 and out["VIS", 0][0] == -1.0
    with pytest.raises(ValueError, match="unknown field name: NOPE"):
        transform(fields, {("NOPE", 0): m0})
    bad = np.zeros(4)
    update_metadata(bad, spin=2)
    with pytest.raises(ValueError, match="spin mismatch for field 'POS': map has spin 2, field has spin 0"):
        transform(fields, {("POS", 0): bad})


def test_configured_datapath_without_a_weight_file_raises(tmp_path):
    """healpy raises when ``use_pixel_weights=True`` finds no file under ``datapath`` (heracles/healpy.py:183-189 passes both):
    a configured data path that holds no file for the resolution must not fall back to unit weights silently."""
    import heracles_amd as hx

    mapper = hx.HipHealpixMapper(8, 12, deconvolve=False, niter=0, datapath=tmp_path)
    with pytest.raises(FileNotFoundError, match="healpix_full_weights_nside_0008.fits"):
        mapper._load_weights()
    with pytest.raises(FileNotFoundError):
        mapper.transform(mapper.create())


def test_result_array_checks_the_callers_out():
    """`out=` of mixmat / mixmat_eb / MixmatContext: the caller's array is used as it is or refused -- the library keeps no pool of
    released results (round 4's reference-count heuristics are gone)."""
    from heracles_amd import _lib

    assert not hasattr(_lib, "_HostPool") and not hasattr(_lib, "host_empty")
    fresh = _lib.result_array((3, 5, 7))
    assert fresh.shape == (3, 5, 7) and fresh.dtype == np.float64 and fresh.flags.c_contiguous
    mine = np.zeros((3, 5, 7))
    assert _lib.result_array((3, 5, 7), mine) is mine
    with pytest.raises(ValueError):
        _lib.result_array((3, 5, 7), np.zeros((3, 5, 8)))
    with pytest.raises(ValueError):
        _lib.result_array((3, 5, 7), np.zeros((3, 5, 7), dtype=np.float32))
    with pytest.raises(ValueError):
        _lib.result_array((5, 7), np.zeros((7, 5)).T)
    ro = np.zeros((2, 2))
    ro.flags.writeable = False
    with pytest.raises(ValueError):
        _lib.result_array((2, 2), ro)
    class Shaped:
        shape = (2, 2)

    with pytest.raises(TypeError):
        _lib.result_array((2, 2), Shaped())


@pytest.mark.parametrize("blk,spin,m,lmax,nside", [
    (32, 2, 2500, 6144, None), (16, 2, 5800, 6144, None), (32, 0, 4000, 6144, None),      # round 5: lmax 6144, a grid of colatitudes
    (32, 2, 5944, 6144, 4096), (32, 0, 5944, 6144, 4096), (16, 2, 5944, 6144, 4096),      # the plan's own rings, m up to lmax - 200
    (32, 2, 7800, 8000, 8192), (32, 0, 7800, 8000, 8192), (16, 2, 7800, 8000, 8192),      # examples/heracles.cfg:56-62: lmax 8000
    (32, 2, 7000, 8000, 4096), (16, 0, 7800, 8000, 4096),
    (32, 2, 12088, 12288, 8192), (32, 0, 12088, 12288, 8192), (16, 2, 12088, 12288, 8192), (16, 0, 10500, 12288, 8192),
    (32, 2, 9000, 12288, 8192), (32, 2, 300, 12288, 8192),
])
def test_dead_block_margins_cover_the_growth_of_the_recursion(blk, spin, m, lmax, nside):
    """k_legendre_duo / k_synth_duo skip the matrix work of a block whose chains all enter it below 2^-(100 + E_b).  The margins E_b in the
    kernels must cover what a skipped chain can reach inside the block (near l = m a step multiplies by up to sqrt(2m / (l - m))): the
    long-double emulation of the normalised recursions (tools/calibrate_dead_blocks.py) gives the margin needed for nothing above 2^-75 of
    a value of lambda to be left out; the kernels keep at least 6 bits on top of it.  Round 6: the margin needed SATURATES with m (the rings
    that matter are those next to the pruning limit, where the growth depends on m / (l sin theta), not on m): 52.2 / 26.3 / 18.1 bits for
    the first three 32-l blocks at lmax 12288, m 12088 against 51.8 / 25.9 / 17.6 at lmax 6144 -- the constants hold for every plan the
    library accepts (nside <= 8192, lmax <= 3 nside / 2); on the rings of the plan itself (nside given) as on a grid of colatitudes.
    Full record: profiles/r06_dead_block_calibration.txt."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("calib", os.path.join(os.path.dirname(__file__), "..", "tools", "calibrate_dead_blocks.py"))
    calib = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(calib)
    needed = calib.need(m, spin, blk=blk, nbmax=14, nth=1500, lmax=lmax, nside=nside)
    if m >= 1000:
        assert needed[0] > 10  # the first block is the one that needs a margin at all
    for b, e in enumerate(needed):
        assert calib.kernel_margin(b, blk) >= e + 6, (b, e)
