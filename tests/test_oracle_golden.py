"""Pins the CPU oracle against golden vectors produced by the reference's own numpy code
(tests/golden/make_golden.py) and against exact known answers (sympy 3j, numpy leggauss)."""

import numpy as np
import pytest

from helpers import key_str

NAMES = [("POS", 0), ("POS", 1), ("SHE", 0), ("SHE", 1)]


def test_alm2lmax(oracle, golden):
    for n, v in zip(golden["alm2lmax_sizes"], golden["alm2lmax_values"]):
        assert oracle.alm2lmax(int(n)) == v


def test_alm2cl_blocks(oracle, golden):
    from itertools import combinations_with_replacement

    for k1, k2 in combinations_with_replacement(NAMES, 2):
        a, b = golden[f"alm/{key_str(k1)}"], golden[f"alm/{key_str(k2)}"]
        ref = golden[f"alm2cl/{key_str(k1)}/{key_str(k2)}"]
        out = oracle.alm2cl(a, b)
        assert out.shape == ref.shape
        np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(oracle.alm2cl(golden["alm/POS|0"]), golden["alm2cl_auto_default/POS|0"], rtol=1e-12)
    for tag, lm in (("lmax20", 20), ("lmax40", 40)):
        ref = golden[f"alm2cl_{tag}/POS|0/SHE|1"]
        out = oracle.alm2cl(golden["alm/POS|0"], golden["alm/SHE|1"], lmax=lm)
        assert out.shape == ref.shape  # lmax beyond the data does not zero-pad
        np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-14)


def test_alm2cl_unequal(oracle, golden):
    a1, a2 = golden["uneq/a1"], golden["uneq/a2"]
    np.testing.assert_allclose(oracle.alm2cl(a1, a2), golden["uneq/cl"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(oracle.alm2cl(a1, a2, lmax=20), golden["uneq/cl_lmax20"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(oracle.alm2cl(a2, a1), golden["uneq/cl_rev"], rtol=1e-12, atol=1e-14)


def test_legendre_funcs(oracle, golden):
    lmax = 40
    for i, x in enumerate(golden["leg/x"]):
        (P, dP), (d20, d22, d2m2) = oracle.legendre_funcs(lmax, float(x))
        np.testing.assert_allclose(P, golden[f"leg/{i}/P"], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(dP, golden[f"leg/{i}/dP"], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(d20, golden[f"leg/{i}/d20"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(d22, golden[f"leg/{i}/d22"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(d2m2, golden[f"leg/{i}/d2m2"], rtol=1e-9, atol=1e-11)


def test_wigner_d_recursion_vs_closed_forms(oracle, golden):
    # the recursion route agrees with the reference's closed forms away from x -> 1
    lmax = 40
    for i, x in enumerate(golden["leg/x"]):
        if x > 0.99:
            continue
        np.testing.assert_allclose(oracle.wigner_d(lmax, 0, 0, float(x)), golden[f"leg/{i}/P"], atol=1e-13)
        np.testing.assert_allclose(oracle.wigner_d(lmax, 2, 0, float(x))[2:], golden[f"leg/{i}/d20"], atol=1e-12)
        np.testing.assert_allclose(oracle.wigner_d(lmax, 2, 2, float(x))[2:], golden[f"leg/{i}/d22"], atol=1e-12)
        np.testing.assert_allclose(oracle.wigner_d(lmax, 2, -2, float(x))[2:], golden[f"leg/{i}/d2m2"], atol=1e-11)
        np.testing.assert_allclose(oracle.wigner_d(lmax, 1, 1, float(x))[1:], golden[f"leg/{i}/d11"], atol=1e-12)
        np.testing.assert_allclose(oracle.wigner_d(lmax, -1, 1, float(x))[1:], golden[f"leg/{i}/dm11"], atol=1e-12)


@pytest.mark.parametrize("lm", [12, 40, 97])
def test_cl2corr_corr2cl(oracle, golden, lm):
    cls = golden[f"c2c/{lm}/cls"]
    corr = oracle.cl2corr(cls)
    np.testing.assert_allclose(corr, golden[f"c2c/{lm}/corr"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(oracle.corr2cl(golden[f"c2c/{lm}/corr"]), golden[f"c2c/{lm}/cls_back"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(oracle.cl2corr(cls[:, 0]), golden[f"c2c/{lm}/corr1d"], rtol=1e-9, atol=1e-12)
    x, w = oracle.gauss_legendre(lm + 1)
    np.testing.assert_allclose(x, golden[f"c2c/{lm}/x"], atol=2e-15)
    np.testing.assert_allclose(w, golden[f"c2c/{lm}/w"], rtol=1e-10)  # numpy leggauss is the less exact side


def test_gauss_legendre_exactness(oracle):
    for n in (1, 2, 5, 64, 513):
        x, w = oracle.gauss_legendre(n)
        assert abs(w.sum() - 2) < 1e-13
        # integrates x^(2n-2) exactly
        k = 2 * n - 2
        assert abs((w * x**k).sum() - 2 / (k + 1)) < 1e-12


def test_wigner3j_vs_sympy(oracle):
    from sympy import N
    from sympy.physics.wigner import wigner_3j

    for l1, l2, m1, m2 in [(2, 2, 2, -2), (5, 3, 2, -2), (6, 6, 0, 0), (6, 6, 2, -2), (7, 4, 0, 0),
                           (12, 12, 2, -2), (10, 2, 2, -2), (3, 8, 2, -2), (4, 4, 0, 0), (0, 0, 0, 0), (2, 7, 0, 0)]:
        jmin, w = oracle.wigner3j_l3(l1, l2, m1, m2)
        for i, v in enumerate(w):
            ref = float(N(wigner_3j(l1, l2, jmin + i, m1, m2, -(m1 + m2)), 30))
            assert abs(ref - v) < 1e-14


def test_mixmat_identities(oracle):
    L = 24
    cl = np.zeros(L + 1)
    cl[0] = 4 * np.pi  # full-sky mask: W_l = 4 pi delta_l0  =>  M = identity
    for s in [(0, 0), (0, 2), (2, 0)]:
        lo = max(abs(s[0]), abs(s[1]))
        M = oracle.mixmat(cl, spin=s)
        np.testing.assert_allclose(M[lo:, lo:], np.eye(L + 1 - lo), atol=1e-13)
    M3 = oracle.mixmat_eb(cl)
    np.testing.assert_allclose(M3[0][2:, 2:], np.eye(L - 1), atol=1e-13)
    np.testing.assert_allclose(M3[1], 0, atol=1e-13)
    # dense quadrature form D^T diag(w xi) D of SURVEY.md 8a-7
    rng = np.random.default_rng(3)
    cl = 1 / (1 + np.arange(L + 1)) ** 2 * rng.uniform(0.5, 1.5, L + 1)
    x, w = oracle.gauss_legendre(3 * L // 2 + 1)
    xi = sum((2 * l + 1) / (4 * np.pi) * cl[l] * np.array([oracle.wigner_d(L, 0, 0, xx)[l] for xx in x]) for l in range(L + 1))

    def G(a, b):
        D = np.array([oracle.wigner_d(L, a, b, xx) for xx in x])
        return (D.T * (w * xi)) @ D * ((2 * np.arange(L + 1) + 1) / 2)[None, :]

    np.testing.assert_allclose(G(0, 0), oracle.mixmat(cl, spin=(0, 0)), atol=1e-14)
    np.testing.assert_allclose(G(2, 0), oracle.mixmat(cl, spin=(2, 0)), atol=1e-14)
    M3 = oracle.mixmat_eb(cl)
    np.testing.assert_allclose((G(2, 2) + G(2, -2)) / 2, M3[0], atol=1e-14)
    np.testing.assert_allclose((G(2, 2) - G(2, -2)) / 2, M3[1], atol=1e-14)
    np.testing.assert_allclose(G(2, -2), M3[2], atol=1e-14)
    # row-sum rule: sum_{l2} M_{0 l2} = sum (2 l3+1) W_l3 / 4pi  when l2max >= l3max
    M = oracle.mixmat(cl, l1max=4, l2max=2 * L, l3max=L)
    assert abs(M[0].sum() - ((2 * np.arange(L + 1) + 1) * cl).sum() / (4 * np.pi)) < 1e-13
    assert oracle.mixmat(cl, l1max=10, l2max=20).shape == (11, 21)


def _racah_3j(j1, j2, j3, m1, m2):
    """Exact Wigner 3j (j1 j2 j3; m1 m2 -(m1+m2)) from Racah's single sum, written as an alternating sum of products of
    three binomials (exact big integers; the sum cancels over thousands of digits at l ~ 6000) and rounded once at the
    end (50 digits): independent of every recursion, usable where sympy needs ~10 s per symbol."""
    from math import comb

    import mpmath

    m3 = -(m1 + m2)
    if j3 < abs(j1 - j2) or j3 > j1 + j2 or abs(m1) > j1 or abs(m2) > j2 or abs(m3) > j3:
        return 0.0
    a, b, c = j1 + j2 - j3, j1 - j2 + j3, -j1 + j2 + j3
    k0 = max(0, j2 - j3 - m1, j1 - j3 + m2)
    k1 = min(a, j1 - m1, j2 + m2)
    # sum_k (-1)^k / (k! (a-k)! (j1-m1-k)! (j3-j2+m1+k)! (j2+m2-k)! (j3-j1-m2+k)!) = S / (a! b! c!)
    S = 0
    A, B, Cc = comb(a, k0), comb(b, j1 - m1 - k0), comb(c, j2 + m2 - k0)
    for k in range(k0, k1 + 1):                  # the three binomials by their exact ratios (math.comb per term: 30 s)
        t = A * B * Cc
        S += -t if k & 1 else t
        if k < k1:
            A = A * (a - k) // (k + 1)
            n = j1 - m1 - k
            B = B * n // (b - n + 1)
            n = j2 + m2 - k
            Cc = Cc * n // (c - n + 1)
    with mpmath.workdps(50):
        F = mpmath.factorial
        sq = (F(j1 + m1) * F(j1 - m1) * F(j2 + m2) * F(j2 - m2) * F(j3 + m3) * F(j3 - m3)) / (F(j1 + j2 + j3 + 1) * F(a) * F(b) * F(c))
        v = mpmath.sqrt(sq) * mpmath.mpf(abs(S))
        sign = (-1 if (j1 - j2 - m3) & 1 else 1) * (-1 if S < 0 else 1)
        return float(sign * v)


def test_racah_sum_agrees_with_sympy():
    from sympy import N
    from sympy.physics.wigner import wigner_3j

    for j1, j2, j3, m1, m2 in [(2, 2, 2, 2, -2), (12, 9, 7, 2, -2), (30, 28, 11, 0, 0), (40, 40, 80, 2, -2), (25, 31, 6, 2, -2),
                               (17, 17, 1, 2, -2), (9, 14, 20, 0, 0)]:
        ref = float(N(wigner_3j(j1, j2, j3, m1, m2, -(m1 + m2)), 30))
        assert abs(_racah_3j(j1, j2, j3, m1, m2) - ref) <= 1e-15 * max(1.0, abs(ref))
    # one symbol at config 4's size, evaluated once with sympy (12 s) and kept as a constant
    assert abs(_racah_3j(4096, 4000, 500, 2, -2) - 0.000388713221742114777088655531457) < 1e-18


def test_wigner3j_recursion_at_config4_and_bench_sizes(oracle):
    """The Schulten-Gordon recursion where the GPU tests lean on it: rows and columns near L = 4096 / 6144, the diagonal, an
    off-diagonal pair and a pair whose l3 range starts in the non-classical tail, against the exact sum."""
    cases = []
    for L in (4096, 6144):
        for l1, l2 in [(L, L), (L, L - 200), (L - 23, L // 2), (L // 2 + 7, L // 2), (L, 2), (L - 5, 37)]:
            lo, hi = abs(l1 - l2), l1 + l2
            for m1, m2 in ((2, -2), (0, 0)):
                picks = sorted({lo, lo + 1, (lo + hi) // 2, (lo + hi) // 2 + 1, min(hi, L) - 1, min(hi, L), hi})
                cases.append((l1, l2, m1, m2, picks))
    worst = 0.0
    for l1, l2, m1, m2, picks in cases:
        jmin, w = oracle.wigner3j_l3(l1, l2, m1, m2)
        scale = np.abs(w).max()
        for l3 in picks:
            if l3 < jmin:
                continue
            ref = _racah_3j(l1, l2, l3, m1, m2)
            worst = max(worst, abs(w[l3 - jmin] - ref) / scale)
    assert worst < 5e-13, worst


def test_mixmat_blocks_vs_sympy_and_full(oracle):
    """hxo_mixmat_block / hxo_mixmat_eb_block: equal to the full matrices on every block, and to exact 3j sums at L = 40 (the big-integer
    Racah sum, which test_racah_sum_agrees_with_sympy pins on sympy)."""
    L = 40
    rng = np.random.default_rng(17)
    cl = rng.uniform(0.5, 1.5, L + 1) / (1 + np.arange(L + 1)) ** 1.5
    full = {s: oracle.mixmat(cl, spin=s) for s in [(0, 0), (2, 0), (0, 2)]}
    eb = oracle.mixmat_eb(cl)
    for rows, cols in [((0, L), (0, L)), ((33, 40), (5, 29)), ((17, 17), (0, 40)), ((2, 9), (38, 40))]:
        r, c = slice(rows[0], rows[1] + 1), slice(cols[0], cols[1] + 1)
        for s, M in full.items():
            np.testing.assert_array_equal(oracle.mixmat_block(cl, rows, cols, spin=s), M[r, c])
        np.testing.assert_array_equal(oracle.mixmat_eb_block(cl, rows, cols), eb[:, r, c])
    # exact sums on a block that touches the corner l1 = l2 = l3max (truncated l3 range)
    rows, cols = (37, 40), (36, 40)
    w3 = {}

    def w(l1, l2, l3, m):
        key = (l1, l2, l3, m)
        if key not in w3:
            w3[key] = _racah_3j(l1, l2, l3, m, -m)
        return w3[key]

    ref00 = np.zeros((4, 5)); ref20 = np.zeros((4, 5)); refeb = np.zeros((3, 4, 5))
    for i, l1 in enumerate(range(rows[0], rows[1] + 1)):
        for j, l2 in enumerate(range(cols[0], cols[1] + 1)):
            f = (2 * l2 + 1) / (4 * np.pi)
            for l3 in range(abs(l1 - l2), L + 1):
                t = (2 * l3 + 1) * cl[l3]
                ref00[i, j] += f * t * w(l1, l2, l3, 0) ** 2
                ref20[i, j] += f * t * w(l1, l2, l3, 2) * w(l1, l2, l3, 0)
                refeb[(l1 + l2 + l3) & 1, i, j] += f * t * w(l1, l2, l3, 2) ** 2
    refeb[2] = refeb[0] - refeb[1]
    np.testing.assert_allclose(oracle.mixmat_block(cl, rows, cols, spin=(0, 0)), ref00, atol=1e-15)
    np.testing.assert_allclose(oracle.mixmat_block(cl, rows, cols, spin=(2, 0)), ref20, atol=1e-15)
    np.testing.assert_allclose(oracle.mixmat_eb_block(cl, rows, cols), refeb, atol=1e-15)
