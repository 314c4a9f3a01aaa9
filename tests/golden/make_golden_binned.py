"""Golden vectors for the binned results of the two-point drivers, from the reference's own functions.

Run ONCE in the build container (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden_binned.py

``heracles.result.binned``, ``heracles.twopoint.angular_power_spectra(bins=, weights=)`` and
``heracles.twopoint.mixing_matrices(bins=, weights=)`` are importable through the bare-package shim of make_golden.py.  The
third-party ``convolvecl`` is absent from this image, so for the mixing-matrix cases the reference's driver is handed the ORACLE's
``mixmat`` / ``mixmat_eb`` (oracle/hxoracle.py: the 3j recursion) in its place: what is pinned there is the reference's binning of full
matrices -- rows, weights, zero rule, angular arrays -- not convolvecl's arithmetic (parity unpinned).
Only inputs and outputs are stored -- no reference source text.
"""

import os
import sys
import types
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import key_str, mock_alms, ref_modules  # noqa: E402


def store(out, tag, res):
    out[f"{tag}/array"] = np.asarray(res.array)
    for name in ("ell", "lower", "upper", "weight"):
        val = getattr(res, name)
        if isinstance(val, tuple):
            for n, v in enumerate(val):
                out[f"{tag}/{name}{n}"] = np.asarray(v)
        else:
            out[f"{tag}/{name}"] = np.asarray(val)
    out[f"{tag}/axis"] = np.array(res.axis)
    md = res.array.dtype.metadata or {}
    out[f"{tag}/md"] = np.array(sorted(f"{a}={md[a]!r}" for a in md))


def main():
    from oracle import hxoracle as ho

    h = ref_modules()
    Result, binned = h.result.Result, h.result.binned
    rng = np.random.default_rng(606)
    out = {}

    # ---- binned() on spectra of every spin shape (result.py:124-248 as called at twopoint.py:283-284) ----
    L = 40
    ell = np.arange(L + 1)
    warr = rng.uniform(0.5, 2.0, L + 11)  # longer than the axis: the reference cuts it
    edges = {
        "lmin2": np.array([2, 5, 10, 20, 41]),          # l < 2 left out
        "short": np.array([0, 3, 9, 27]),               # l >= 27 left out
        "beyond": np.array([1.5, 4.5, 30.0, 60, 100]),  # non-integer edges, one empty bin at the end
        "log": np.unique(np.geomspace(2, L + 1, 9).astype(int)),
    }
    weights = {"none": None, "ll1": "l(l+1)", "2l1": "2l+1", "arr": warr}
    shapes = {"s00": ((), (0, 0)), "s02": ((2,), (0, 2)), "s22": ((2, 2), (2, 2))}
    for en, ed in edges.items():
        out[f"cl/edges/{en}"] = ed
    out["cl/warr"] = warr
    for sn, (shp, spin) in shapes.items():
        arr = rng.standard_normal(shp + (L + 1,)) / (1 + ell) ** 1.5
        arr[..., 7] = 0.0  # an exact zero inside a bin
        arr.dtype = np.dtype(arr.dtype, metadata={"spin_1": spin[0], "spin_2": spin[1], "bias": 0.25})
        out[f"cl/in/{sn}"] = np.asarray(arr)
        for en, ed in edges.items():
            for wn, w in weights.items():
                store(out, f"cl/{sn}/{en}/{wn}", binned(Result(arr, spin=spin, axis=-1), ed, w))
    # all-zero rows and a result that is binned already (its own ell / weight arrays enter the rule)
    z = np.zeros((2, L + 1))
    z[1] = rng.standard_normal(L + 1)
    out["cl/in/zero"] = z
    store(out, "cl/zero", binned(Result(z, spin=(0, 2), axis=-1), edges["lmin2"], "2l+1"))
    pre = Result(rng.standard_normal(12), spin=(0, 0), axis=-1, ell=np.linspace(3, 58, 12), weight=rng.uniform(1, 9, 12))
    out["cl/in/pre"], out["cl/in/pre_ell"], out["cl/in/pre_weight"] = pre.array, pre.ell, pre.weight
    out["cl/edges/pre"] = np.array([0, 10, 30, 59, 80])
    store(out, "cl/pre/none", binned(pre, out["cl/edges/pre"]))
    store(out, "cl/pre/ll1", binned(pre, out["cl/edges/pre"], "l(l+1)"))
    # a bare array (no Result): last axis
    store(out, "cl/bare", binned(out["cl/in/s02"], edges["lmin2"], "2l+1"))

    # ---- binned() on matrices, axis = -2 (twopoint.py:391-397), rectangular ----
    n, m = 31, 51
    mat = rng.standard_normal((n, m))
    mat3 = rng.standard_normal((3, n, m))
    out["mat/in"], out["mat/in3"] = mat, mat3
    out["mat/edges"] = np.array([2, 4, 8, 16, 31])
    out["mat/warr"] = rng.uniform(0.5, 2.0, n)
    for wn, w in (("none", None), ("2l1", "2l+1"), ("arr", out["mat/warr"])):
        store(out, f"mat/one/{wn}", binned(Result(mat, spin=(0, 2), axis=-2, ell=np.arange(n)), out["mat/edges"], w))
        store(out, f"mat/three/{wn}", binned(Result(mat3, spin=(2, 2), axis=-2, ell=np.arange(n)), out["mat/edges"], w))
    # two angular axes, one set of edges per axis
    two = Result(mat, spin=(0, 0), axis=(0, 1), ell=(np.arange(n), np.arange(m)))
    e2 = (np.array([0, 5, 11, 31]), np.array([1, 7, 20, 40, 51]))
    out["mat/two/edges0"], out["mat/two/edges1"] = e2
    store(out, "mat/two", binned(two, e2, ("2l+1", None)))

    # ---- angular_power_spectra(bins=, weights=) (twopoint.py:173-299) ----
    alms = mock_alms(np.random.default_rng(50))
    alms_b = {}
    for (nm, i), a in alms.items():
        b = np.array(a)
        md = dict(a.dtype.metadata)
        if i == 0:
            md.update(fsky=0.5, musq=1.2, dens=3.4)
        md.update(geometry="plain", kernel="plain")
        b.dtype = np.dtype(b.dtype, metadata=md)
        alms_b[nm, i] = b
        out[f"aps/alm/{key_str((nm, i))}"] = np.asarray(b)
    out["aps/edges"] = np.array([2, 4, 8, 16, 33])
    for wn, w in (("none", None), ("2l1", "2l+1"), ("ll1", "l(l+1)")):
        cls = h.twopoint.angular_power_spectra(alms_b, bins=out["aps/edges"], weights=w)
        out[f"aps/{wn}/keys"] = np.array([key_str(k) for k in cls])
        for k, v in cls.items():
            store(out, f"aps/{wn}/{key_str(k)}", v)

    # ---- mixing_matrices(bins=, weights=): the reference's driver around the oracle's matrices ----
    conv = types.ModuleType("convolvecl")
    conv.mixmat, conv.mixmat_eb = ho.mixmat, ho.mixmat_eb
    fields = {
        "POS": types.SimpleNamespace(mask="VIS", spin=0),
        "SHE": types.SimpleNamespace(mask="WHT", spin=2),
    }
    cases = {
        # rectangular, l1max != l2max as in examples/heracles.cfg (2000 / 4000 there)
        "rect": dict(l1max=48, l2max=96, l3max=144, edges=np.unique(np.geomspace(2, 49, 9).astype(int))),
        "square": dict(l1max=40, l2max=40, l3max=80, edges=np.array([0, 3, 10, 22, 41])),
        "short": dict(l1max=36, l2max=30, l3max=60, edges=np.array([5, 9, 20, 30])),   # rows below 5 and from 30 on left out
    }
    for cn, c in cases.items():
        l3 = c["l3max"]
        l = np.arange(l3 + 1)
        mcls = {}
        for n_, key in enumerate((("VIS", "VIS", 0, 1), ("VIS", "WHT", 0, 1), ("WHT", "WHT", 0, 1))):
            mcls[key] = 4 * np.pi * 0.3 * np.exp(-l * (l + 1) / (300.0 + 100 * n_)) + 1e-3 / (1 + l) ** 2
            out[f"mm/{cn}/mcl/{key_str(key)}"] = mcls[key]
        out[f"mm/{cn}/edges"] = c["edges"]
        out[f"mm/{cn}/lmax"] = np.array([c["l1max"], c["l2max"], c["l3max"]])
        warr = rng.uniform(0.5, 2.0, c["l1max"] + 1)
        out[f"mm/{cn}/warr"] = warr
        for wn, w in (("none", None), ("2l1", "2l+1"), ("arr", warr)):
            with mock.patch.dict(sys.modules, {"convolvecl": conv}):
                mms = h.twopoint.mixing_matrices(fields, mcls, l1max=c["l1max"], l2max=c["l2max"], l3max=l3, bins=c["edges"], weights=w)
            out[f"mm/{cn}/{wn}/keys"] = np.array([key_str(k) for k in mms])
            for k, v in mms.items():
                store(out, f"mm/{cn}/{wn}/{key_str(k)}", v)

    np.savez_compressed(os.path.join(HERE, "reference_binned.npz"), **out)
    print("wrote", len(out), "arrays to reference_binned.npz")


if __name__ == "__main__":
    main()
