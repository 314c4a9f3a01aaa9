"""Generate golden vectors for the host-side numpy part of the hot path.

Run ONCE in the build container (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden.py

The reference package cannot be imported as a whole here (`fitsio` is absent, an
ordinary ModuleNotFoundError), so its numpy/scipy-only modules are imported through a
bare package object.  Only inputs and outputs are stored -- no
reference source text.  Third-party arithmetic (healpy, convolvecl, ducc0) is absent,
so map2alm and mixmat values are NOT produced here ("parity unpinned").
"""

import importlib
import os
import sys
import types
from unittest import mock

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def ref_modules():
    pkg = types.ModuleType("heracles")
    pkg.__path__ = [os.path.join(REF, "heracles")]
    sys.modules["heracles"] = pkg
    mods = {}
    for name in ("core", "result", "progress", "utils", "twopoint", "transforms", "unmixing"):
        mods[name] = importlib.import_module(f"heracles.{name}")
    return types.SimpleNamespace(**mods)


def mock_alms(rng, lmax=32):
    """Same construction as the reference fixture tests/test_twopoint.py:22-39."""
    size = (lmax + 1) * (lmax + 2) // 2
    fields = {"POS": 0, "SHE": 2}
    zbins = (0, 1)
    alms = {}
    for n, s in fields.items():
        shape = (size, 2) if s == 0 else (2, size, 2)
        for i in zbins:
            a = rng.standard_normal(shape) @ [1, 1j]
            a.dtype = np.dtype(a.dtype, metadata={"nside": 32, "spin": s})
            alms[n, i] = a
    return alms


def key_str(key):
    return "|".join(str(k) for k in key)


def main():
    h = ref_modules()
    rng = np.random.default_rng(50)
    out = {}

    # ---- alm2lmax / alm2cl (twopoint.py:55-101) --------------------------------------
    alms = mock_alms(rng)
    names = list(alms)
    for k, a in alms.items():
        out[f"alm/{key_str(k)}"] = np.asarray(a)
    from itertools import combinations_with_replacement

    for k1, k2 in combinations_with_replacement(names, 2):
        out[f"alm2cl/{key_str(k1)}/{key_str(k2)}"] = h.twopoint.alm2cl(alms[k1], alms[k2])
    out["alm2cl_auto_default/POS|0"] = h.twopoint.alm2cl(alms["POS", 0])
    out["alm2cl_lmax20/POS|0/SHE|1"] = h.twopoint.alm2cl(alms["POS", 0], alms["SHE", 1], lmax=20)
    out["alm2cl_lmax40/POS|0/SHE|1"] = h.twopoint.alm2cl(alms["POS", 0], alms["SHE", 1], lmax=40)
    # unequal sizes (tests/test_twopoint.py:68-88)
    l1, l2 = 10, 20
    a1 = rng.standard_normal(((l1 + 1) * (l1 + 2) // 2, 2)) @ [1, 1j]
    a2 = rng.standard_normal(((l2 + 1) * (l2 + 2) // 2, 2)) @ [1, 1j]
    out["uneq/a1"], out["uneq/a2"] = a1, a2
    out["uneq/cl"] = h.twopoint.alm2cl(a1, a2)
    out["uneq/cl_lmax20"] = h.twopoint.alm2cl(a1, a2, lmax=l2)
    out["uneq/cl_rev"] = h.twopoint.alm2cl(a2, a1)
    out["alm2lmax_sizes"] = np.array([(l + 1) * (l + 2) // 2 for l in range(0, 1000, 7)])
    out["alm2lmax_values"] = np.array(
        [h.twopoint.alm2lmax(np.zeros(n)) for n in out["alm2lmax_sizes"]]
    )

    # ---- angular_power_spectra (twopoint.py:173-299) ---------------------------------
    fsky, musq, dens = 0.5, 1.2, 3.4
    alms_b = {}
    for (n, i), a in alms.items():
        b = np.array(a)
        md = dict(a.dtype.metadata)
        if i == 0:
            md.update(fsky=fsky, musq=musq, dens=dens)
        md.update(geometry="plain", kernel="plain")
        b.dtype = np.dtype(b.dtype, metadata=md)
        alms_b[n, i] = b
    for tag, kw in (
        ("plain", {}),
        ("nodebias", {"debias": False}),
        ("lmax16", {"lmax": 16}),
        ("incl", {"include": [("POS", "SHE", ..., ...)]}),
        ("excl", {"exclude": [("SHE", "SHE")]}),
    ):
        cls = h.twopoint.angular_power_spectra(alms_b, **kw)
        out[f"aps/{tag}/keys"] = np.array([key_str(k) for k in cls])
        for k, v in cls.items():
            out[f"aps/{tag}/cl/{key_str(k)}"] = np.asarray(v.array)
            md = v.array.dtype.metadata or {}
            out[f"aps/{tag}/md/{key_str(k)}"] = np.array(
                sorted(f"{a}={md[a]!r}" for a in md)
            )
    # two separate sets, to exercise the key canonicalisation / swap
    a1s = {k: v for k, v in alms_b.items() if k[1] == 0}
    a2s = {k: v for k, v in alms_b.items() if k[1] == 1}
    cls = h.twopoint.angular_power_spectra(a1s, a2s)
    out["aps/cross/keys"] = np.array([key_str(k) for k in cls])
    for k, v in cls.items():
        out[f"aps/cross/cl/{key_str(k)}"] = np.asarray(v.array)
    # reversed insertion order: second-seen name order gets swapped
    rev = dict(reversed(list(alms_b.items())))
    cls = h.twopoint.angular_power_spectra(rev)
    out["aps/rev/keys"] = np.array([key_str(k) for k in cls])
    for k, v in cls.items():
        out[f"aps/rev/cl/{key_str(k)}"] = np.asarray(v.array)

    # ---- _debias_cl, non-healpix kernels (twopoint.py:104-170) -----------------------
    dcases = {
        "a": (np.zeros(100), 1.23, {}),
        "c": (np.zeros((2, 100)), None, {"bias": 4.56, "spin_2": 2}),
        "d": (np.zeros((2, 2, 3, 100)), 7.89, {"spin_1": 2, "spin_2": 2}),
        "e": (np.zeros((2, 2, 3, 100)), 7.89, {"spin_1": 0, "spin_2": 0}),
    }
    for k, (arr, bias, md) in dcases.items():
        arr = arr + rng.standard_normal(arr.shape)
        arr.dtype = np.dtype(arr.dtype, metadata=md)
        out[f"debias/{k}/in"] = np.asarray(arr)
        out[f"debias/{k}/out"] = np.asarray(h.twopoint._debias_cl(arr, bias))

    # ---- legendre_funcs / _cl2corr / _corr2cl (transforms.py:46-204) -----------------
    lmax = 40
    xs = np.array([-0.93, -0.2, 0.0, 0.31, 0.9, 0.9985, 0.99995, 0.9999999])
    out["leg/x"] = xs
    for i, x in enumerate(xs):
        (P, dP), (d11, dm11), (d20, d22, d2m2) = h.transforms.legendre_funcs(lmax, x, m=(0, 1, 2))
        out[f"leg/{i}/P"], out[f"leg/{i}/dP"] = P, dP
        out[f"leg/{i}/d11"], out[f"leg/{i}/dm11"] = d11, dm11
        out[f"leg/{i}/d20"], out[f"leg/{i}/d22"], out[f"leg/{i}/d2m2"] = d20, d22, d2m2
    for lm in (12, 40, 97):
        cls4 = rng.standard_normal((lm + 1, 4)) / (1 + np.arange(lm + 1))[:, None] ** 2
        out[f"c2c/{lm}/cls"] = cls4
        corr = h.transforms._cl2corr(cls4)
        out[f"c2c/{lm}/corr"] = corr
        out[f"c2c/{lm}/cls_back"] = h.transforms._corr2cl(corr)
        out[f"c2c/{lm}/corr1d"] = h.transforms._cl2corr(cls4[:, 0])
        xv, wv = h.transforms._cached_gauss_legendre(lm + 1)
        out[f"c2c/{lm}/x"], out[f"c2c/{lm}/w"] = xv, wv

    # ---- dict-level cl2corr / corr2cl / naturalspice (transforms.py:207-363,
    #      unmixing.py:36-102) on Result objects ---------------------------------------
    Result = h.result.Result
    L = 24
    ell = np.arange(L + 1)
    shapes = {("POS", "POS", 0, 0): (), ("POS", "SHE", 0, 0): (2,), ("SHE", "SHE", 0, 0): (2, 2)}
    spins = {("POS", "POS", 0, 0): (0, 0), ("POS", "SHE", 0, 0): (0, 2), ("SHE", "SHE", 0, 0): (2, 2)}
    d = {}
    for k, shp in shapes.items():
        arr = rng.standard_normal(shp + (L + 1,)) / (1 + ell) ** 2
        if spins[k][0] or spins[k][1]:
            arr[..., :2] = 0.0
        d[k] = Result(arr, spin=spins[k], axis=-1, ell=ell)
        out[f"dict/d/{key_str(k)}"] = arr
    wd = h.transforms.cl2corr(d)
    back = h.transforms.corr2cl(wd)
    for k in d:
        out[f"dict/wd/{key_str(k)}"] = np.asarray(wd[k].array)
        out[f"dict/back/{key_str(k)}"] = np.asarray(back[k].array)
    # mask cls for naturalspice: smooth positive spectra, lmax_mask = 2 L
    Lm = 2 * L
    ellm = np.arange(Lm + 1)
    mcl = {}
    for k in (("VIS", "VIS", 0, 0), ("VIS", "WHT", 0, 0), ("WHT", "WHT", 0, 0)):
        arr = 4 * np.pi * 0.4 * np.exp(-ellm * (ellm + 1) / 200.0) + 1e-4 / (1 + ellm) ** 2
        arr = arr * rng.uniform(0.9, 1.1)
        mcl[k] = Result(arr, spin=(0, 0), axis=-1, ell=ellm)
        out[f"ns/m/{key_str(k)}"] = arr
    fields = {
        "POS": types.SimpleNamespace(mask="VIS", spin=0),
        "SHE": types.SimpleNamespace(mask="WHT", spin=2),
    }
    for tag, tm in (("default", None), ("theta30", 30.0)):
        # naturalspice mutates the mask correlation functions in place -> fresh copies
        mcl_c = {k: Result(np.array(v.array), spin=v.spin, axis=-1, ell=ellm) for k, v in mcl.items()}
        res = h.unmixing.naturalspice(d, mcl_c, fields, theta_max=tm)
        for k, v in res.items():
            out[f"ns/{tag}/{key_str(k)}"] = np.asarray(v.array)

    # ---- mixing_matrices driver logic with the third-party kernel mocked -------------
    # (tests/test_twopoint.py:293-383): record which (spin, which-function) calls are made
    calls = []

    def fake(name):
        def f(cl, l1max=None, l2max=None, l3max=None, spin=None):
            calls.append((name, tuple(spin), l1max, l2max, l3max))
            n = len(cl)
            return np.zeros((n, n)) if name == "mixmat" else np.zeros((3, n, n))

        return f

    conv = types.ModuleType("convolvecl")
    conv.mixmat = fake("mixmat")
    conv.mixmat_eb = fake("mixmat_eb")
    with mock.patch.dict(sys.modules, {"convolvecl": conv}):
        cl = rng.standard_normal(21)
        mm_cls = {
            ("VIS", "VIS", 0, 1): cl, ("VIS", "WHT", 0, 1): cl, ("WHT", "VIS", 0, 1): cl,
            ("WHT", "WHT", 0, 1): cl, ("X", "Y", 0, 1): cl, ("WHT", "WHT", 1, 1): cl,
        }
        mflds = {
            "POS": types.SimpleNamespace(mask="VIS", spin=0),
            "SHE": types.SimpleNamespace(mask="WHT", spin=2),
            "POS2": types.SimpleNamespace(mask="VIS", spin=0),
            "NOMASK": types.SimpleNamespace(mask=None, spin=0),
        }
        mms = h.twopoint.mixing_matrices(mflds, mm_cls, l1max=10, l2max=12, l3max=20)
    out["mm/keys"] = np.array([key_str(k) for k in mms])
    out["mm/calls"] = np.array([repr(c) for c in calls])
    out["mm/shapes"] = np.array([repr(np.shape(v.array)) for v in mms.values()])
    out["mm/axis"] = np.array([repr(v.axis) for v in mms.values()])

    np.savez_compressed(os.path.join(OUT, "reference_numpy.npz"), **out)
    print(f"wrote {len(out)} arrays to reference_numpy.npz")


if __name__ == "__main__":
    main()
