"""GPU parity: all-pairs alm x alm -> Cl (hx_alm2cl_pairs through the C ABI) against the
oracle and the reference's golden vectors."""

from itertools import combinations_with_replacement

import numpy as np
import pytest

from helpers import key_str, random_alm

pytestmark = pytest.mark.gpu

NAMES = [("POS", 0), ("POS", 1), ("SHE", 0), ("SHE", 1)]
RTOL = 1e-12  # f64 sums of <= lmax terms in a different order than the reference's running mean


def test_alm2cl_golden(golden):
    import heracles_amd as hx

    for k1, k2 in combinations_with_replacement(NAMES, 2):
        a, b = golden[f"alm/{key_str(k1)}"], golden[f"alm/{key_str(k2)}"]
        ref = golden[f"alm2cl/{key_str(k1)}/{key_str(k2)}"]
        out = hx.alm2cl(a, b)
        assert out.shape == ref.shape
        np.testing.assert_allclose(out, ref, rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(hx.alm2cl(golden["alm/POS|0"]), golden["alm2cl_auto_default/POS|0"], rtol=RTOL)
    for tag, lm in (("lmax20", 20), ("lmax40", 40)):
        ref = golden[f"alm2cl_{tag}/POS|0/SHE|1"]
        out = hx.alm2cl(golden["alm/POS|0"], golden["alm/SHE|1"], lmax=lm)
        assert out.shape == ref.shape
        np.testing.assert_allclose(out, ref, rtol=RTOL, atol=1e-14)
    a1, a2 = golden["uneq/a1"], golden["uneq/a2"]
    np.testing.assert_allclose(hx.alm2cl(a1, a2), golden["uneq/cl"], rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(hx.alm2cl(a1, a2, lmax=20), golden["uneq/cl_lmax20"], rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(hx.alm2cl(a2, a1), golden["uneq/cl_rev"], rtol=RTOL, atol=1e-14)


@pytest.mark.parametrize("lmax", [0, 1, 63, 64, 65, 200])
def test_alm2cl_vs_oracle_ragged(oracle, lmax):
    import heracles_amd as hx

    rng = np.random.default_rng(lmax)
    a = random_alm(rng, lmax, 0, (3,))
    b = random_alm(rng, lmax + 5, 0, (2,))
    # imaginary parts of m=0 must be ignored (twopoint.py:88)
    a[..., : lmax + 1] += 1j * rng.standard_normal((3, lmax + 1))
    out = hx.alm2cl(a, b)
    ref = oracle.alm2cl(a, b)
    assert out.shape == (3, 2, lmax + 1)
    np.testing.assert_allclose(out, ref, rtol=RTOL, atol=1e-13)


def test_alm2cl_pairs_many_components(oracle):
    import heracles_amd as hx

    rng = np.random.default_rng(11)
    lmax = 97
    comps = [random_alm(rng, lmax) for _ in range(11)]
    pairs = [(i, j) for i in range(11) for j in range(i, 11)] + [(3, 1), (0, 0)]
    out = hx.alm2cl_pairs(comps, pairs, lmax)
    for n, (i, j) in enumerate(pairs):
        np.testing.assert_allclose(out[n], oracle.alm2cl(comps[i], comps[j]), rtol=RTOL, atol=1e-13)
    # run-to-run bitwise repeatability (fixed-order reduction)
    out2 = hx.alm2cl_pairs(comps, pairs, lmax)
    np.testing.assert_array_equal(out, out2)


def test_alm2cl_full_size_properties():
    """lmax = 6144 (BASELINE metric size): size-independent properties instead of an oracle run."""
    import heracles_amd as hx

    rng = np.random.default_rng(5)
    lmax = 6144
    nlm = (lmax + 1) * (lmax + 2) // 2
    a = rng.standard_normal(nlm) + 1j * rng.standard_normal(nlm)
    b = rng.standard_normal(nlm) + 1j * rng.standard_normal(nlm)
    caa, cab, cbb, cba = hx.alm2cl_pairs([a, b], [(0, 0), (0, 1), (1, 1), (1, 0)], lmax)
    np.testing.assert_array_equal(cab, cba)                       # symmetry, bitwise
    cs = hx.alm2cl_pairs([a + b], [(0, 0)], lmax)[0]               # bilinearity
    np.testing.assert_allclose(cs, caa + 2 * cab + cbb, rtol=1e-10, atol=1e-12)
    assert (caa > 0).all()
    # expectation: <|a|^2> = 2 per mode for m>0
    assert abs(caa[1000:].mean() - 2.0) < 0.02
    # exact last multipole from the definition
    l = lmax
    idx = np.array([m * (2 * lmax + 1 - m) // 2 + l for m in range(l + 1)])
    ref = (a[idx[0]].real * b[idx[0]].real + 2 * (a[idx[1:]] * np.conj(b[idx[1:]])).real.sum()) / (2 * l + 1)
    assert abs(cab[l] - ref) < 1e-12 * max(1, abs(ref))


def test_mixed_lmax_in_one_call(oracle):
    """Alms of different band limits in one angular_power_spectra call (the reference's alm2cl supports it,
    twopoint.py:78-99): every output-lmax group sweeps a tile at ITS lmax, so components with a smaller
    band limit that share the component list / a tile must never be indexed outside their triangle."""
    import heracles_amd as hx

    rng = np.random.default_rng(2024)
    lm = {"A": 50, "B": 1000, "C": 100, "D": 7}
    alms = {}
    for n, (name, lmax) in enumerate(lm.items()):
        spin = 2 if name in ("B", "D") else 0
        a = random_alm(rng, lmax, spin, (2,) if spin else ())
        a.dtype = np.dtype(a.dtype, metadata={"spin": spin, "nside": 64})
        alms[name, 0] = a
    cls = hx.angular_power_spectra(alms, debias=False)
    assert len(cls) == 10
    for (k1, k2, i1, i2), res in cls.items():
        ref = oracle.alm2cl(np.asarray(alms[k1, i1]), np.asarray(alms[k2, i2]))
        assert res.array.shape == ref.shape, (k1, k2)
        np.testing.assert_allclose(np.asarray(res.array), ref, rtol=RTOL, atol=1e-13)
    # the raw kernel with a low-lmax component inside a tile swept at a much higher lmax_out
    comps = [random_alm(rng, 1000), random_alm(rng, 50), random_alm(rng, 1000), random_alm(rng, 3)]
    out = hx.alm2cl_pairs(comps, [(0, 2), (2, 0), (0, 0)], 1000)
    np.testing.assert_allclose(out[0], oracle.alm2cl(comps[0], comps[2]), rtol=RTOL, atol=1e-13)
    np.testing.assert_array_equal(out[0], out[1])
    with pytest.raises(hx.HxError):
        hx.alm2cl_pairs(comps, [(0, 1)], 1000)  # a REQUESTED pair beyond its band limit is an error


def test_m_range_partial_sums_add_up(oracle):
    """hx_alm2cl_pairs_range: the spectra of disjoint m-ranges add up to the full sum (the m-sharded route's all-reduce)."""
    import heracles_amd as hx

    rng = np.random.default_rng(321)
    lmax = 70
    nlm = (lmax + 1) * (lmax + 2) // 2
    comps = [rng.standard_normal((nlm, 2)) @ [1, 1j] for _ in range(5)]
    pairs = [(i, j) for i in range(5) for j in range(i, 5)]
    full = hx.alm2cl_pairs(comps, pairs, lmax)
    bounds = [0, 1, 9, 40, lmax + 1]
    parts = [hx.alm2cl_pairs(comps, pairs, lmax, m_range=(bounds[q], bounds[q + 1])) for q in range(4)]
    np.testing.assert_allclose(sum(parts), full, rtol=1e-13, atol=1e-15)
    # the first range is m = 0 alone: Re a Re b / (2l + 1)
    np.testing.assert_allclose(parts[0][3], comps[0][: lmax + 1].real * comps[3][: lmax + 1].real / (2 * np.arange(lmax + 1) + 1), rtol=1e-14)
    assert not hx.alm2cl_pairs(comps, pairs, lmax, m_range=(5, 5)).any()
    # strided sets (the orders of rank q of 3: q, q + 3, ...) add up as well
    strided = [hx.alm2cl_pairs(comps, pairs, lmax, m_range=(q, lmax + 1, 3)) for q in range(3)]
    np.testing.assert_allclose(sum(strided), full, rtol=1e-13, atol=1e-15)


def test_pairs_in_either_orientation_and_duplicates(oracle):
    """A pair list that names component pairs as (high, low), (low, high), on the diagonal blocks both ways, and some twice -- the
    order a buffer with its spin-2 components in front of its spin-0 ones produces (heracles_amd.distributed): every row is the
    spectrum of ITS pair, and (a, b) equals (b, a) bit for bit."""
    import heracles_amd as hx

    lmax, ncomp = 90, 17
    rng = np.random.default_rng(23)
    alms = [random_alm(rng, lmax) for _ in range(ncomp)]
    pairs = [(a, b) for a in range(ncomp) for b in range(ncomp) if (a * 7 + b * 3) % 4 != 1]
    pairs += [(16, 0), (0, 16), (16, 0), (5, 5), (5, 5), (11, 2)]
    out = hx.alm2cl_pairs(alms, pairs, lmax)
    assert out.shape == (len(pairs), lmax + 1)
    first = {}
    for k, (a, b) in enumerate(pairs):
        key = (min(a, b), max(a, b))
        if key in first:
            np.testing.assert_array_equal(out[k], out[first[key]])
        else:
            first[key] = k
            np.testing.assert_allclose(out[k], oracle.alm2cl(alms[a], alms[b]), rtol=RTOL, atol=1e-14)


@pytest.mark.gpu
def test_alm2cl_pairs_into_the_callers_array():
    """out=: the spectra land in the caller's array (a page-locked one in a loop), which is what the call returns; the wrong shape is refused"""
    import heracles_amd as hx
    from heracles_amd.twopoint import alm2cl_pairs

    lmax = 40
    nlm = (lmax + 1) * (lmax + 2) // 2
    rng = np.random.default_rng(11)
    comps = [rng.standard_normal(nlm) + 1j * rng.standard_normal(nlm) for _ in range(3)]
    pairs = [(0, 0), (0, 1), (1, 2), (2, 2)]
    ref = alm2cl_pairs(comps, pairs, lmax)
    for out in (np.full((4, lmax + 1), np.nan), hx.pinned_empty((4, lmax + 1))):
        got = alm2cl_pairs(comps, pairs, lmax, out=out)
        assert got is out
        np.testing.assert_array_equal(got, ref)
    with pytest.raises(ValueError):
        alm2cl_pairs(comps, pairs, lmax, out=np.empty((3, lmax + 1)))
