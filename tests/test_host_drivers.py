"""The host drivers that post-process spectra -- ``invert_mixing_matrix`` / ``apply_mixing_matrix`` (heracles/twopoint.py:404-524),
dict-level ``cl2corr`` / ``corr2cl`` and ``naturalspice`` (heracles/transforms.py:207-363, heracles/unmixing.py:36-102), ``debias_cls``
(heracles/twopoint.py:302-313) -- against vectors generated from the reference, WITHOUT a GPU: their arithmetic kernels (hx_pinv,
hx_matvec, hx_cl2corr / hx_corr2cl, hx_gauss_legendre) are replaced by numpy / the oracle as checker stubs, so what is tested is
the Python layer: keys, spin cases, angular arrays, the in-place damping quirk.  The same functions run with the kernels in place in
tests/test_gpu_mixmat.py and tests/test_gpu_widen.py.  This file also runs under tests/test_production_config.py with the reference's
own Result / TocDict / update_metadata in place of heracles_amd's (``heracles_amd.core.HAVE_HERACLES``)."""

import os
import types

import numpy as np
import pytest

import heracles_amd as hx
from heracles_amd import core, transforms as tr, twopoint as tp, unmixing as um
from helpers import key_str

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_which_core_classes_are_in_use():
    """HAVE_HERACLES follows whether ``heracles.core`` / ``heracles.result`` are importable: the reference's classes then replace ours"""
    shim = os.environ.get("HX_TEST_HERACLES_SHIM") == "1"
    print(f"HAVE_HERACLES={core.HAVE_HERACLES}")
    assert core.HAVE_HERACLES is shim
    if shim:
        import heracles.core
        import heracles.result

        assert hx.Result is heracles.result.Result and core.Result is heracles.result.Result
        assert hx.TocDict is heracles.core.TocDict and hx.toc_match is heracles.core.toc_match
        assert hx.update_metadata is heracles.core.update_metadata
        assert heracles.result.__file__.startswith("/root/reference/")
    else:
        assert hx.Result.__module__ == "heracles_amd.core"


@pytest.fixture
def host_kernels(monkeypatch, oracle):
    def batch(fn, specs, lmax):
        f = oracle.cl2corr if fn.__name__ == "hx_cl2corr" else oracle.corr2cl
        return np.stack([f(s, lmax) for s in specs])

    monkeypatch.setattr(tr, "_batch", batch)
    monkeypatch.setattr(tr, "gauss_legendre", oracle.gauss_legendre)
    monkeypatch.setattr(um, "gauss_legendre", oracle.gauss_legendre)
    monkeypatch.setattr(tp, "pinv", lambda M, rcond=1e-5: np.linalg.pinv(np.asarray(M), rcond=rcond))
    monkeypatch.setattr(tp, "_matvec", lambda M, xs: np.atleast_2d(xs) @ np.asarray(M).T)
    return oracle


def test_dict_transforms_and_naturalspice(host_kernels, golden):
    L = 24
    ell = np.arange(L + 1)
    keys = {("POS", "POS", 0, 0): (0, 0), ("POS", "SHE", 0, 0): (0, 2), ("SHE", "SHE", 0, 0): (2, 2)}
    d = {k: hx.Result(np.array(golden[f"dict/d/{key_str(k)}"]), spin=s, axis=-1, ell=ell) for k, s in keys.items()}
    wd = hx.cl2corr(d)
    back = hx.corr2cl(wd)
    for k in d:
        np.testing.assert_allclose(wd[k].array, golden[f"dict/wd/{key_str(k)}"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(back[k].array, golden[f"dict/back/{key_str(k)}"], rtol=1e-8, atol=1e-12)
        assert type(wd[k]) is hx.Result and wd[k].spin == keys[k]
    ellm = np.arange(2 * L + 1)
    fields = {"POS": types.SimpleNamespace(mask="VIS", spin=0), "SHE": types.SimpleNamespace(mask="WHT", spin=2)}
    for tag, tm in (("default", None), ("theta30", 30.0)):
        m = {k: hx.Result(np.array(golden[f"ns/m/{key_str(k)}"]), spin=(0, 0), axis=-1, ell=ellm)
             for k in (("VIS", "VIS", 0, 0), ("VIS", "WHT", 0, 0), ("WHT", "WHT", 0, 0))}
        res = hx.naturalspice(d, m, fields, theta_max=tm)
        for k in d:
            ref = golden[f"ns/{tag}/{key_str(k)}"]
            np.testing.assert_allclose(res[k].array, ref, rtol=1e-6, atol=1e-9 * np.abs(ref).max())
            np.testing.assert_array_equal(res[k].ell, ell)


def test_invert_and_apply_mixing_matrix(host_kernels):
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_mixing.npz"))
    spins = {"POS|POS|0|0": (0, 0), "POS|SHE|0|1": (0, 2), "SHE|SHE|1|1": (2, 2)}

    def key_of(ks):
        return tuple(int(x) if x.isdigit() else x for x in ks.split("|"))

    for name in ("square", "tall", "wide"):
        mats, rconds, cls = {}, {}, {}
        for ks, sp in spins.items():
            M = g[f"{name}/M/{ks}"]
            mats[key_of(ks)] = hx.Result(M, spin=sp, axis=-2, ell=np.arange(M.shape[-2]))
            rconds[key_of(ks)] = float(g[f"{name}/rcond/{ks}"])
            cls[key_of(ks)] = hx.Result(g[f"{name}/cl/{ks}"], spin=sp, axis=-1)
        inv = hx.invert_mixing_matrix(mats, rcond=rconds)
        assert list(inv) == list(mats)
        for ks in spins:
            ref = g[f"{name}/inv/{ks}"]
            got = inv[key_of(ks)]
            assert type(got) is hx.Result and got.array.shape == ref.shape and got.spin == mats[key_of(ks)].spin
            np.testing.assert_allclose(got.array, ref, atol=1e-11 * np.abs(ref).max())
            np.testing.assert_array_equal(got.ell, g[f"{name}/inv_ell/{ks}"])
        applied = hx.apply_mixing_matrix(cls, inv)
        for ks in spins:
            ref = g[f"{name}/applied/{ks}"]
            np.testing.assert_allclose(applied[key_of(ks)].array, ref, atol=1e-11 * np.abs(ref).max())
        with pytest.raises(KeyError, match="Missing rcond value"):
            hx.invert_mixing_matrix(mats, rcond={})


def test_debias_cls_over_a_toc_dict(golden):
    cls = hx.TocDict()
    for k in ("a", "c", "d", "e"):
        arr = np.array(golden[f"debias/{k}/in"])
        md = {"a": {}, "c": {"bias": 4.56, "spin_2": 2}, "d": {"spin_1": 2, "spin_2": 2}, "e": {"spin_1": 0, "spin_2": 0}}[k]
        arr.dtype = np.dtype(arr.dtype, metadata=md)
        cls["F", "F", k, k] = arr
    bias = {("F", "F", "a", "a"): 1.23, ("F", "F", "d", "d"): 7.89, ("F", "F", "e", "e"): 7.89}
    out = hx.debias_cls(cls, bias)
    assert type(out) is hx.TocDict and list(out) == list(cls)
    for k in ("a", "c", "d", "e"):
        np.testing.assert_allclose(out["F", "F", k, k], golden[f"debias/{k}/out"], rtol=1e-14, atol=1e-15)
        np.testing.assert_array_equal(cls["F", "F", k, k], golden[f"debias/{k}/in"])  # not in place
    assert out["F"].keys() == out.keys()  # TocDict selection by prefix
