"""Device-resident jackknife loop (heracles/dices/jackknife.py:41-248) against a CPU restatement that does
what the reference does step by step with the oracle: region maps, transforms, full - sum(regions), all-pairs Cl, bias
and footprint corrections."""

import types
from itertools import combinations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NSIDE, LMAX, NJK = 16, 24, 4


def _setup(rng):
    import heracles_amd as hx

    npix = 12 * NSIDE**2
    mapper = hx.HipHealpixMapper(NSIDE, LMAX, deconvolve=False, niter=0)
    fields = {"POS": types.SimpleNamespace(spin=0, mapper_or_error=mapper, mask="VIS"),
              "SHE": types.SimpleNamespace(spin=2, mapper_or_error=mapper, mask="WHT")}
    jk = np.zeros(npix)
    # four longitude wedges inside a footprint; pixels outside the footprint (label 0) belong to no region
    theta = np.arccos(1 - 2 * (np.arange(npix) + 0.5) / npix)
    footprint = theta < 2.0
    jk[footprint] = 1 + (np.arange(npix)[footprint] % NJK)
    maps = {}
    for name, bins in (("POS", (1, 2)), ("SHE", (1,))):
        for b in bins:
            m = rng.standard_normal(((2,) if name == "SHE" else ()) + (npix,)) * footprint
            md = {"spin": fields[name].spin, "nside": NSIDE, "kernel": "healpix", "fsky": 0.4, "musq": 1.3 + b, "dens": 2.5}
            m = np.ascontiguousarray(m)
            m.dtype = np.dtype(m.dtype, metadata=md)
            maps[name, b] = m
    return fields, maps, jk


def _expected(oracle, fields, maps, jk, regions, unmixed=False):
    from heracles_amd.jackknife import jackknife_fsky

    alms = {}
    for key, m in maps.items():
        spin = fields[key[0]].spin
        m2 = np.asarray(m).reshape(-1, m.shape[-1])
        a = oracle.map2alm(m2, NSIDE, LMAX, spin=spin)
        for r in regions:
            a = a - oracle.map2alm(m2 * (jk == r), NSIDE, LMAX, spin=spin)
        alms[key] = a if spin else a[0]
    f_ratio = jackknife_fsky(jk, *regions)
    f_fast = jackknife_fsky(jk, *regions, ratio=not unmixed)
    exp = {}
    keys = list(maps)
    for n1, k1 in enumerate(keys):
        for k2 in keys[n1:]:
            cl = oracle.alm2cl(alms[k1], alms[k2])
            s1, s2 = fields[k1[0]].spin, fields[k2[0]].spin
            bias = 0
            if k1 == k2:
                md = maps[k1].dtype.metadata
                bias = (0.5 if s1 == 2 else 1.0) * md["fsky"] * md["musq"] / md["dens"]
                lmin = max(s1, s2)
                if s1 == 2:
                    cl[0, 0, lmin:] -= bias
                    cl[1, 1, lmin:] -= bias
                else:
                    cl[..., lmin:] -= bias
            exp[k1[0], k2[0], k1[1], k2[1]] = (cl + bias - bias * f_ratio) / f_fast
    return exp


@pytest.mark.parametrize("nd", [1, 2])
def test_jackknife_cls_fast_correction(oracle, nd):
    import heracles_amd as hx

    rng = np.random.default_rng(77)
    fields, maps, jk = _setup(rng)
    out = hx.jackknife_cls(maps, None, jk, fields, mask_correction="Fast", nd=nd)
    combos = list(combinations(range(1, NJK + 1), nd))
    assert list(out) == combos
    for regions in combos:
        exp = _expected(oracle, fields, maps, jk, regions)
        got = out[regions]
        assert set(got) == set(exp)
        for key, ref in exp.items():
            np.testing.assert_allclose(np.asarray(got[key].array), ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())
        # the bias recorded with the spectrum is the bias of the reduced footprint
        b = got["POS", "POS", 1, 1].array.dtype.metadata["bias"]
        assert np.isclose(b, 0.4 * 2.3 / 2.5 * hx.jackknife.jackknife_fsky(jk, *regions))


def test_region_alms_stay_on_device_and_nd0(oracle):
    import heracles_amd as hx

    rng = np.random.default_rng(78)
    fields, maps, jk = _setup(rng)
    ra = hx.region_alms(fields, maps, jk)
    assert ra.tensor.is_cuda and ra.tensor.shape == (NJK + 1, 4, (LMAX + 1) * (LMAX + 2) // 2)
    # the regions partition the footprint: their alms add up to the alms of the footprint-masked (= full) maps
    total = ra.tensor[1:].sum(dim=0)
    assert float((total - ra.tensor[0]).abs().max()) <= 1e-12 * float(ra.tensor[0].abs().max())
    full = ra.full()
    ref = oracle.map2alm(np.asarray(maps["SHE", 1]), NSIDE, LMAX, spin=2)
    np.testing.assert_allclose(full["SHE", 1].tensor.cpu().numpy(), ref, atol=1e-11 * np.abs(ref).max())
    cls0 = hx.jackknife_cls(maps, None, jk, fields, nd=0)[()]
    direct = hx.angular_power_spectra({k: v.numpy() for k, v in full.items()})
    assert list(cls0) == list(direct)
    for k in cls0:
        np.testing.assert_array_equal(np.asarray(cls0[k].array), np.asarray(direct[k].array))


def test_jackknife_full_mask_correction_runs(oracle):
    """The "Full" correction (mask correlation functions through cl2corr / corr2cl) against the same steps on host arrays."""
    import heracles_amd as hx
    from heracles_amd.jackknife import correct_bias, correct_footprint_naturalspice

    rng = np.random.default_rng(79)
    fields, maps, jk = _setup(rng)
    vis = {}
    for (name, b), m in maps.items():
        v = (jk > 0).astype(float) * (1.0 + 0.1 * b)
        v = np.ascontiguousarray(np.stack([v, 0 * v]) if name == "SHE" else v)
        v.dtype = np.dtype(v.dtype, metadata={"spin": fields[name].spin, "nside": NSIDE})
        vis[name, b] = v
    # mask keys follow the field's mask name in the reference; here data and visibility share their keys
    fields_m = {k: types.SimpleNamespace(spin=f.spin, mapper_or_error=f.mapper_or_error, mask=k) for k, f in fields.items()}
    out = hx.jackknife_cls(maps, vis, jk, fields_m, mask_correction="Full", nd=1)
    regions = (2,)
    d, v = hx.region_alms(fields_m, maps, jk), hx.region_alms(fields_m, vis, jk)
    cls = correct_bias(hx.angular_power_spectra({k: a.numpy() for k, a in d.delete(regions).items()}), jk, *regions)
    mm = hx.angular_power_spectra({k: a.numpy() for k, a in v.delete(regions).items()})
    m0 = hx.angular_power_spectra({k: a.numpy() for k, a in v.full().items()})
    ref = correct_footprint_naturalspice(cls, mm, m0, fields_m)
    for key in ref:
        np.testing.assert_allclose(np.asarray(out[regions][key].array), np.asarray(ref[key].array), rtol=1e-9, atol=1e-12)


def test_full_footprint_correction_against_reference_golden():
    """correct_footprint_naturalspice against golden vectors made by the reference's own cl2corr / _naturalspice / corr2cl /
    binned, composed as heracles/dices/jackknife.py:411-450 composes them (tests/golden/make_golden_jackknife.py)."""
    import os

    import heracles_amd as hx
    from heracles_amd.core import Result
    from heracles_amd.jackknife import correct_footprint_naturalspice

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_jackknife.npz"))
    ks = lambda key: "|".join(str(k) for k in key)  # noqa: E731
    spins = {("POS", "POS", 0, 0): (0, 0), ("POS", "SHE", 0, 0): (0, 2), ("SHE", "SHE", 0, 0): (2, 2)}
    mkeys = (("VIS", "VIS", 0, 0), ("VIS", "WHT", 0, 0), ("WHT", "WHT", 0, 0))
    fields = {"POS": types.SimpleNamespace(mask="VIS", spin=0), "SHE": types.SimpleNamespace(mask="WHT", spin=2)}
    for tag, unmixed in (("mixed", False), ("unmixed", True)):
        cls = {k: Result(np.array(g[f"cls/{ks(k)}"]), spin=s, axis=-1, ell=np.arange(g[f"cls/{ks(k)}"].shape[-1])) for k, s in spins.items()}
        mls0 = {k: Result(np.array(g[f"mls0/{ks(k)}"]), spin=(0, 0), axis=-1, ell=np.arange(g[f"mls0/{ks(k)}"].shape[-1])) for k in mkeys}
        mljk = {k: Result(np.array(g[f"mljk/{ks(k)}"]), spin=(0, 0), axis=-1, ell=np.arange(g[f"mljk/{ks(k)}"].shape[-1])) for k in mkeys}
        got = correct_footprint_naturalspice(cls, mljk, mls0, fields, unmixed=unmixed)
        assert list(got) == list(spins)
        for k in spins:
            ref = g[f"{tag}/out/{ks(k)}"]
            # (the division by the regularised mask correlation amplifies rounding: measured 4.6e-13 absolute on values of 3e-2)
            np.testing.assert_allclose(np.asarray(got[k].array), ref, rtol=1e-7, atol=1e-10 * np.abs(ref).max())
    assert hx is not None


def test_region_alms_use_each_fields_own_mapper(oracle):
    """Two spin-0 fields whose mappers differ in `deconvolve` must not share one batched transform
    (heracles/dices/jackknife.py:143-148 transforms every field with its own mapper)."""
    import heracles_amd as hx

    rng = np.random.default_rng(80)
    npix = 12 * NSIDE**2
    pw = (np.linspace(1.0, 0.7, LMAX + 1), np.linspace(1.0, 0.6, LMAX + 1))
    m_plain = hx.HipHealpixMapper(NSIDE, LMAX, deconvolve=False, niter=0)
    m_dec = hx.HipHealpixMapper(NSIDE, LMAX, deconvolve=True, niter=0, pixwin=pw)
    fields = {"A": types.SimpleNamespace(spin=0, mapper_or_error=m_plain, mask=None),
              "B": types.SimpleNamespace(spin=0, mapper_or_error=m_dec, mask=None),
              "C": types.SimpleNamespace(spin=0, mapper_or_error=m_plain, mask=None)}
    jk = 1.0 + (np.arange(npix) % 2)
    maps = {}
    for name in ("A", "B", "C"):
        m = rng.standard_normal(npix)
        m.dtype = np.dtype(m.dtype, metadata={"spin": 0, "nside": NSIDE})
        maps[name, 0] = m
    ra = hx.region_alms(fields, maps, jk)
    full = ra.full()
    for name in ("A", "B", "C"):
        ref = oracle.map2alm(np.asarray(maps[name, 0])[None], NSIDE, LMAX, spin=0)[0]
        if name == "B":
            fl = 1.0 / pw[0]
            ref = ref * np.concatenate([fl[m:] for m in range(LMAX + 1)])
        np.testing.assert_allclose(full[name, 0].numpy(), ref, atol=1e-11 * np.abs(ref).max())
        assert full[name, 0].dtype.metadata["deconv"] == (name == "B")


def test_region_alms_load_the_datapath_weights_like_transform(oracle, tmp_path):
    """A mapper with a configured data path weights its jackknife alms exactly as ``mapper.transform`` does, whichever of the
    two runs first (``region_alms`` read ``mp.pixel_weights`` before anything had loaded the file; the reference
    transforms every jackknife map through the same ``mapper.transform``, heracles/dices/jackknife.py:143-148)."""
    import heracles_amd as hx
    from heracles_amd import weights as hw

    rng = np.random.default_rng(81)
    npix = 12 * NSIDE**2
    comp = 1e-2 * rng.standard_normal(hw.compressed_size(NSIDE))
    (tmp_path / "full_weights").mkdir()
    hw.write_compressed_weights(tmp_path / "full_weights" / hw.weights_filename(NSIDE), NSIDE, comp)
    mapper = hx.HipHealpixMapper(NSIDE, LMAX, deconvolve=False, niter=0, datapath=tmp_path)  # has not transformed anything yet
    assert mapper.pixel_weights is None
    fields = {"A": types.SimpleNamespace(spin=0, mapper_or_error=mapper, mask=None),
              "S": types.SimpleNamespace(spin=2, mapper_or_error=mapper, mask=None)}
    a = rng.standard_normal(npix)
    a.dtype = np.dtype(a.dtype, metadata={"spin": 0, "nside": NSIDE})
    s = rng.standard_normal((2, npix))
    s.dtype = np.dtype(s.dtype, metadata={"spin": 2, "nside": NSIDE})
    jk = 1.0 + (np.arange(npix) % 3)
    full = hx.region_alms(fields, {("A", 0): a, ("S", 0): s}, jk).full()
    assert mapper.pixel_weights is not None
    np.testing.assert_array_equal(full["A", 0].numpy(), np.asarray(mapper.transform(a, spin=0)))
    np.testing.assert_array_equal(full["S", 0].numpy(), np.asarray(mapper.transform(s, spin=2)))
    ref = oracle.map2alm(np.asarray(a)[None], NSIDE, LMAX, spin=0, pix_weights=oracle.expand_full_weights(NSIDE, comp))[0]
    np.testing.assert_allclose(full["A", 0].numpy(), ref, atol=1e-11 * np.abs(ref).max())
