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
