"""Binned results of the two-point drivers against vectors generated from the reference's own ``heracles.result.binned``,
``angular_power_spectra(bins=, weights=)`` and ``mixing_matrices(bins=, weights=)`` (tests/golden/make_golden_binned.py ->
reference_binned.npz; heracles/result.py:124-248, heracles/twopoint.py:283-284, :391-397).

CPU part: the host rule (``heracles_amd.binning``) and the driver logic with the oracle's matrices as ``context=``.
GPU part: the binned rows built directly on the device (hx_mixctx_set_bins / hx_mixctx_apply_binned).
Tolerance: rtol 1e-11 of every value plus 1e-13 of the largest (sums of <= 6145 products taken in another order).
"""

import os
import types

import numpy as np
import pytest

import heracles_amd as hx
from heracles_amd import binning

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_binned.npz"), allow_pickle=False)


def key_str(key):
    return "|".join(str(k) for k in key)


def check(gold, tag, res, rtol=1e-11):
    want = gold[f"{tag}/array"]
    got = np.asarray(res.array)
    assert got.shape == want.shape, tag
    np.testing.assert_allclose(got, want, rtol=rtol, atol=1e-13 * max(np.abs(want).max(), 1e-300), err_msg=tag)
    assert tuple(res.axis) == tuple(gold[f"{tag}/axis"]), tag
    for name in ("ell", "lower", "upper", "weight"):
        val = getattr(res, name)
        if isinstance(val, tuple):
            for n, v in enumerate(val):
                np.testing.assert_allclose(v, gold[f"{tag}/{name}{n}"], rtol=1e-13, err_msg=f"{tag} {name}{n}")
        else:
            assert f"{tag}/{name}" in gold, f"{tag}: {name} should be a tuple"
            np.testing.assert_allclose(val, gold[f"{tag}/{name}"], rtol=1e-13, err_msg=f"{tag} {name}")
    md = got.dtype.metadata or {}
    assert sorted(f"{a}={md[a]!r}" for a in md) == list(gold[f"{tag}/md"]), tag


SHAPES = {"s00": (0, 0), "s02": (0, 2), "s22": (2, 2)}
WEIGHTS = {"none": None, "ll1": "l(l+1)", "2l1": "2l+1", "arr": "cl/warr"}


@pytest.mark.parametrize("sn", list(SHAPES))
@pytest.mark.parametrize("en", ["lmin2", "short", "beyond", "log"])
def test_binned_spectra(gold, sn, en):
    spin = SHAPES[sn]
    arr = np.array(gold[f"cl/in/{sn}"])
    arr.dtype = np.dtype(arr.dtype, metadata={"spin_1": spin[0], "spin_2": spin[1], "bias": 0.25})
    for wn, w in WEIGHTS.items():
        w = gold[w] if wn == "arr" else w
        res = binning.binned(hx.Result(arr, spin=spin, axis=-1), gold[f"cl/edges/{en}"], w)
        check(gold, f"cl/{sn}/{en}/{wn}", res)
        assert res.spin == spin


def test_binned_zero_rule_prebinned_and_bare(gold):
    check(gold, "cl/zero", binning.binned(hx.Result(gold["cl/in/zero"], spin=(0, 2), axis=-1), gold["cl/edges/lmin2"], "2l+1"))
    assert np.all(np.asarray(binning.binned(hx.Result(gold["cl/in/zero"], spin=(0, 2), axis=-1), gold["cl/edges/lmin2"]).array)[0] == 0.0)
    pre = hx.Result(gold["cl/in/pre"], spin=(0, 0), axis=-1, ell=gold["cl/in/pre_ell"], weight=gold["cl/in/pre_weight"])
    check(gold, "cl/pre/none", binning.binned(pre, gold["cl/edges/pre"]))
    check(gold, "cl/pre/ll1", binning.binned(pre, gold["cl/edges/pre"], "l(l+1)"))
    bare = np.array(gold["cl/in/s02"])  # (no Result: the last axis; the array's own metadata travels)
    bare.dtype = np.dtype(bare.dtype, metadata={"spin_1": 0, "spin_2": 2, "bias": 0.25})
    check(gold, "cl/bare", binning.binned(bare, gold["cl/edges/lmin2"], "2l+1"))
    with pytest.raises(ValueError, match="unknown weights string"):
        binning.binned(pre, gold["cl/edges/pre"], "l")
    with pytest.raises(ValueError, match="different number of ell axes"):
        binning.binned(pre, (gold["cl/edges/pre"],) * 2)
    # a mapping is binned value by value
    both = binning.binned({"a": pre, "b": pre}, gold["cl/edges/pre"])
    assert set(both) == {"a", "b"}
    check(gold, "cl/pre/none", both["b"])


def test_binned_matrices_rows_and_two_axes(gold):
    n = gold["mat/in"].shape[0]
    for wn, w in (("none", None), ("2l1", "2l+1"), ("arr", gold["mat/warr"])):
        check(gold, f"mat/one/{wn}", binning.binned(hx.Result(gold["mat/in"], spin=(0, 2), axis=-2, ell=np.arange(n)), gold["mat/edges"], w))
        check(gold, f"mat/three/{wn}", binning.binned(hx.Result(gold["mat/in3"], spin=(2, 2), axis=-2, ell=np.arange(n)), gold["mat/edges"], w))
    m = gold["mat/in"].shape[1]
    two = hx.Result(gold["mat/in"], spin=(0, 0), axis=(0, 1), ell=(np.arange(n), np.arange(m)))
    check(gold, "mat/two", binning.binned(two, (gold["mat/two/edges0"], gold["mat/two/edges1"]), ("2l+1", None)))


FIELDS = {
    "POS": types.SimpleNamespace(mask="VIS", spin=0),
    "SHE": types.SimpleNamespace(mask="WHT", spin=2),
}
MM_KEYS = (("VIS", "VIS", 0, 1), ("VIS", "WHT", 0, 1), ("WHT", "WHT", 0, 1))


def mm_case(gold, cn):
    l1, l2, l3 = (int(v) for v in gold[f"mm/{cn}/lmax"])
    mcls = {key: gold[f"mm/{cn}/mcl/{key_str(key)}"] for key in MM_KEYS}
    return l1, l2, l3, mcls


@pytest.mark.parametrize("cn", ["rect", "square", "short"])
def test_mixing_matrices_bins_host_path(gold, oracle, cn):
    """driver logic + host binning of full matrices (a ``context=`` callable: here the oracle's 3j matrices, as in the generator)"""
    l1, l2, l3, mcls = mm_case(gold, cn)

    def ctx(cl, a, b, c, spin):
        return (oracle.mixmat_eb if all(spin) else oracle.mixmat)(cl, l1max=a, l2max=b, l3max=c, spin=spin)

    for wn, w in (("none", None), ("2l1", "2l+1"), ("arr", gold[f"mm/{cn}/warr"])):
        mms = hx.mixing_matrices(FIELDS, mcls, l1max=l1, l2max=l2, l3max=l3, bins=gold[f"mm/{cn}/edges"], weights=w, context=ctx)
        assert [key_str(k) for k in mms] == list(gold[f"mm/{cn}/{wn}/keys"])
        for k, v in mms.items():
            check(gold, f"mm/{cn}/{wn}/{key_str(k)}", v)
            assert v.spin == (FIELDS[k[0]].spin, FIELDS[k[1]].spin)


def test_split_requests_is_a_partition_by_cost():
    from heracles_amd.twopoint import mixing_requests, request_cost, split_requests

    fields = {f"F{n}": types.SimpleNamespace(mask=f"M{n % 3}", spin=2 if n % 2 else 0) for n in range(6)}
    cls = {(f"M{a}", f"M{b}", i, j): None for a in range(3) for b in range(a, 3) for i in range(3) for j in range(i, 3)}
    todo = mixing_requests(fields, cls)
    for world in (1, 2, 3, 8):
        shares = [split_requests(todo, r, world) for r in range(world)]
        assert sorted(sum(shares, []), key=todo.index) == todo  # a partition, order kept within each share
        loads = [sum(request_cost(req[2]) for req in sh) for sh in shares]
        assert max(loads) - min(loads) <= 2, loads
    with pytest.raises(ValueError):
        split_requests(todo, 2, 2)


# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("cn", ["rect", "square", "short"])
def test_gpu_mixing_matrices_bins_vs_reference_binned(gold, cn):
    """the binned rows built on the GPU against the reference's ``binned`` of the oracle's full matrices"""
    l1, l2, l3, mcls = mm_case(gold, cn)
    for wn, w in (("none", None), ("2l1", "2l+1"), ("arr", gold[f"mm/{cn}/warr"])):
        mms = hx.mixing_matrices(FIELDS, mcls, l1max=l1, l2max=l2, l3max=l3, bins=gold[f"mm/{cn}/edges"], weights=w)
        assert [key_str(k) for k in mms] == list(gold[f"mm/{cn}/{wn}/keys"])
        for k, v in mms.items():
            check(gold, f"mm/{cn}/{wn}/{key_str(k)}", v)


@pytest.mark.gpu
@pytest.mark.parametrize("nbins", [1, 7, 32, 100])
def test_gpu_binned_rows_equal_binned_full_matrix(nbins):
    """hx_mixctx_apply_binned == host binning of the GPU's own full matrices, all three kinds, rectangular, any number of bins"""
    l1, l2, l3 = 300, 411, 500
    l = np.arange(l3 + 1)
    cl = 4 * np.pi * 0.3 * np.exp(-l * (l + 1) / 900.0) + 1e-3 / (1 + l) ** 2
    edges = np.unique(np.geomspace(2, l1 + 1, nbins + 1).astype(int)) if nbins > 1 else np.array([10, 200])
    with hx.MixmatContext(l1, l2, l3) as ctx:
        for weights in (None, "2l+1", np.random.default_rng(3).uniform(0.1, 3.0, l1 + 1)):
            plan = binning.BinPlan(np.arange(l1 + 1), edges, weights)
            ctx.set_bins(plan)
            for spin in ((0, 0), (0, 2), (2, 0), (2, 2)):
                full = ctx(cl, spin)
                want = plan.apply(full, full.ndim - 2)
                got = ctx.binned(cl, spin)
                assert got.shape == want.shape
                np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-13 * np.abs(want).max(), err_msg=f"{spin} {nbins}")
                again = ctx.binned(cl, spin)
                assert np.array_equal(got, again), "not bitwise repeatable"


@pytest.mark.gpu
def test_gpu_binned_rows_at_config4_size():
    """L = 4096 (BASELINE configs[3]), the configuration file's `32 log 2l+1` bins: device-built rows == binned(full GPU matrix)"""
    L = 4096
    l = np.arange(L + 1)
    cl = 4 * np.pi * 0.35 * np.exp(-l * (l + 1) / 4.0e4) + 1e-4 / (1 + l) ** 2
    edges = np.unique(np.geomspace(2, L + 1, 33).astype(int))
    fields = {"POS": types.SimpleNamespace(mask="VIS", spin=0), "SHE": types.SimpleNamespace(mask="VIS", spin=2)}
    mms = hx.mixing_matrices(fields, {("VIS", "VIS", 0, 0): cl}, l1max=L, l2max=L, l3max=L, bins=edges, weights="2l+1")
    assert list(mms) == [("POS", "POS", 0, 0), ("POS", "SHE", 0, 0), ("SHE", "SHE", 0, 0)]
    plan = binning.BinPlan(l, edges, "2l+1")
    with hx.MixmatContext(L, L, L) as ctx:
        for key, spin in ((("POS", "POS", 0, 0), (0, 0)), (("POS", "SHE", 0, 0), (0, 2)), (("SHE", "SHE", 0, 0), (2, 2))):
            full = ctx(cl, spin, out=ctx.result_buffer(spin))
            want = plan.apply(full, full.ndim - 2)
            got = np.asarray(mms[key].array)
            assert got.shape == want.shape == ((3,) if all(spin) else ()) + (edges.size - 1, L + 1)
            np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-13 * np.abs(want).max(), err_msg=str(key))
            np.testing.assert_allclose(mms[key].ell, plan.ell, rtol=1e-14)
            assert mms[key].axis == (got.ndim - 2,)


@pytest.mark.gpu
def test_gpu_angular_power_spectra_bins(gold):
    alms = {}
    for name, spin in (("POS", 0), ("SHE", 2)):
        for i in (0, 1):
            a = np.array(gold[f"aps/alm/{name}|{i}"])
            md = {"nside": 32, "spin": spin, "geometry": "plain", "kernel": "plain"}
            if i == 0:
                md.update(fsky=0.5, musq=1.2, dens=3.4)
            a.dtype = np.dtype(a.dtype, metadata=md)
            alms[name, i] = a
    for wn, w in (("none", None), ("2l1", "2l+1"), ("ll1", "l(l+1)")):
        cls = hx.angular_power_spectra(alms, bins=gold["aps/edges"], weights=w)
        assert [key_str(k) for k in cls] == list(gold[f"aps/{wn}/keys"])
        for k, v in cls.items():
            check(gold, f"aps/{wn}/{key_str(k)}", v, rtol=1e-10)


@pytest.mark.gpu
def test_gpu_release_caches_between_stages():
    """hx_release_caches hands back what the one-shot mixing-matrix entry points and hx_alm2cl_pairs keep in HBM between calls (ADVICE r5:
    none of it is counted by a plan); the next call builds the same result again"""
    import torch

    l = np.arange(301)
    cl = 1.0 / (1.0 + l) ** 2
    a = hx.mixmat_eb(cl)
    rng = np.random.default_rng(1)
    alm = rng.standard_normal((2, 51 * 52 // 2, 2)) @ [1, 1j]
    c1 = hx.alm2cl(alm[0], alm[1])
    torch.cuda.synchronize()
    before = torch.cuda.mem_get_info()[0]
    hx.release_caches()
    after = torch.cuda.mem_get_info()[0]
    assert after >= before  # (tables of L = 300 and the staging buffer are small, but they are gone)
    np.testing.assert_array_equal(hx.mixmat_eb(cl), a)
    np.testing.assert_array_equal(hx.alm2cl(alm[0], alm[1]), c1)
    hx.mixmat_release()
    hx.release_caches()  # (twice in a row: nothing left to free)


@pytest.mark.gpu
def test_gpu_binned_rows_at_bench_size_against_the_3j_oracle(oracle):
    """The binned rows of the L = 6144 matrices (the bench's size, the configuration file's log bins) against the 3j ORACLE, not against
    this library's own full matrix: for three bins -- the first, one in the middle, and the last and widest (1 240 rows up to l = 6144) -- the
    oracle's block of those rows at a handful of columns (on the band, beside it, far from it) is binned on the host and compared with the
    row the GPU built from its binned Wigner-d tables: spins (0,0), (0,2) and the three spin-2 x spin-2 matrices."""
    L = 6144
    l = np.arange(L + 1)
    wl = 4 * np.pi * 0.35 * np.exp(-l * (l + 1) / 3000.0) + 1e-3 / (1.0 + l) ** 2
    edges = np.unique(np.geomspace(2, L + 1, 33).astype(int))
    plan = binning.BinPlan(l, edges, "2l+1")
    picks = [0, plan.nbins // 2, plan.nbins - 1]
    with hx.MixmatContext(L, L, L) as ctx:
        ctx.set_bins(plan)
        got = {spin: ctx.binned(wl, spin) for spin in ((0, 0), (0, 2), (2, 2))}
    for b in picks:
        rows = np.flatnonzero(plan.which == b)
        lo, hi = int(rows[0]), int(rows[-1])
        mid = (lo + hi) // 2
        cols = sorted({max(mid - 1, 0), mid, min(hi + 40, L), min(2 * mid + 7, L), L // 3, L})
        w = plan.w[lo : hi + 1]
        for c in cols:
            ref00 = oracle.mixmat_block(wl, (lo, hi), (c, c), spin=(0, 0))[:, 0]
            ref02 = oracle.mixmat_block(wl, (lo, hi), (c, c), spin=(0, 2))[:, 0]
            refeb = oracle.mixmat_eb_block(wl, (lo, hi), (c, c))[:, :, 0]
            for spin, ref in (((0, 0), ref00), ((0, 2), ref02)):
                want = (w * ref).sum() / plan.norm[b]
                scale = np.abs(got[spin]).max()
                assert abs(got[spin][b, c] - want) <= 1e-12 * scale, (spin, b, c, got[spin][b, c], want)
            scale = np.abs(got[2, 2]).max()
            for k in range(3):
                want = (w * refeb[k]).sum() / plan.norm[b]
                assert abs(got[2, 2][k, b, c] - want) <= 1e-12 * scale, ("eb", k, b, c, got[2, 2][k, b, c], want)


def _config_bins(lmin, lmax, n=32):
    """the edges ``heracles.cli.bins_from_config`` makes of ``bins = 32 log 2l+1`` (heracles/cli.py:356-362): FLOAT edges, exact ends"""
    arr = 10 ** np.linspace(np.log10(lmin), np.log10(lmax + 1), n + 1)
    arr[0], arr[-1] = lmin, lmax + 1
    return arr


@pytest.mark.gpu
@pytest.mark.parametrize("name,fields,l1max,l2max", [
    ("clustering", {"D": ("V", 0)}, 2000, 4000),
    ("shear", {"G": ("W", 2)}, 3000, 5000),
    ("ggl", {"D": ("V", 0), "G": ("W", 2)}, 1000, 2000),
])
def test_gpu_example_configuration_sections(name, fields, l1max, l2max):
    """The three [spectra:*] sections of the reference's example configuration (examples/heracles.cfg:4-25: lmin 10, ``bins = 32 log 2l+1``,
    rectangular lmax / l2max, no l3max: the mask spectra set it -- here 6000, the visibility's lmax) as ``heracles/cli.py:696-716`` runs them:
    float bin edges, rows below lmin left out.  Binned rows built on the GPU == the host rule on the GPU's own full matrices; angular
    arrays as the reference defines them."""
    l3max = 6000
    l = np.arange(l3max + 1)
    flds = {k: types.SimpleNamespace(mask=m, spin=s) for k, (m, s) in fields.items()}
    masks = sorted({m for m, _ in fields.values()})
    mcls = {}
    for i, a in enumerate(masks):
        for b in masks[i:]:
            mcls[a, b, 0, 1] = 4 * np.pi * 0.3 * np.exp(-l * (l + 1) / (2.0e4 + 3e3 * len(mcls))) + 1e-4 / (1 + l) ** 2
    edges = _config_bins(10, l1max)
    assert edges.size == 33 and edges[1] != round(edges[1])
    mms = hx.mixing_matrices(flds, mcls, l1max=l1max, l2max=l2max, l3max=None, bins=edges, weights="2l+1")
    plan = binning.BinPlan(np.arange(l1max + 1), edges, "2l+1")
    assert plan.which[:10].max() == -1 and plan.which[10] == 0 and plan.which[-1] == 31
    assert len(mms) >= 1
    with hx.MixmatContext(l1max, l2max, l3max) as ctx:
        for key, res in mms.items():
            spin = (flds[key[0]].spin, flds[key[1]].spin)
            mkey = next(k for k in mcls if {k[0], k[1]} == {flds[key[0]].mask, flds[key[1]].mask} or (k[0] == flds[key[0]].mask and k[1] == flds[key[1]].mask))
            full = ctx(mcls[mkey], spin)
            want = plan.apply(full, full.ndim - 2)
            got = np.asarray(res.array)
            assert got.shape == ((3,) if all(spin) else ()) + (32, l2max + 1)
            np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-13 * np.abs(want).max(), err_msg=f"{name} {key}")
            np.testing.assert_array_equal(res.lower, edges[:-1])
            np.testing.assert_array_equal(res.upper, edges[1:])
            np.testing.assert_allclose(res.weight, plan.norm, rtol=1e-14)
            assert res.axis == (got.ndim - 2,) and tuple(res.spin) == spin
