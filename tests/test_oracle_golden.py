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
