"""GPU parity: Gauss-Legendre nodes, Wigner-d tables, mixing matrices and Cl <-> xi
transforms through the C ABI, against the oracle (3j recursion, pinned by sympy) and the
reference's golden vectors."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_gauss_legendre(oracle):
    import heracles_amd as hx

    for n in (1, 2, 3, 64, 97, 1000, 4097):
        x, w = hx.gauss_legendre(n)
        xo, wo = oracle.gauss_legendre(n)
        np.testing.assert_allclose(x, xo, atol=3e-16 * 4)
        np.testing.assert_allclose(w, wo, rtol=2e-12)  # (the weights belong to the nodes x + xlo: evaluated at the rounded x the end-point ones are 6e-10 off)
        assert abs(w.sum() - 2) < 1e-12


def test_gauss_legendre_double_double_nodes(oracle):
    """hx_gauss_legendre_dd: node k = x[k] + xlo[k].  x is the oracle's node (long-double Newton, rounded) to an ulp, |xlo| stays below
    an ulp of 1, and P_n(x + xlo) vanishes to first order: P_n(x) + xlo P_n'(x) ~ 1e-16 x the size P_n(x) has at a node that is only
    known to a double (checked in long double)."""
    import ctypes  # noqa: F401

    import heracles_amd as hx
    from heracles_amd import _lib

    hx.init()
    for n in (7, 155, 1000, 6145):
        x, w, xlo = np.empty(n), np.empty(n), np.empty(n)
        _lib.check(_lib.load().hx_gauss_legendre_dd(n, _lib.ptr(x), _lib.ptr(w), _lib.ptr(xlo)))
        xo, wo = oracle.gauss_legendre(n)
        np.testing.assert_allclose(x, xo, atol=2.3e-16)
        assert np.abs(xlo).max() <= 2.3e-16
        np.testing.assert_allclose(w, wo, rtol=2e-12)
        ld = np.longdouble
        t = x.astype(ld)
        p0, p1 = np.ones_like(t), t.copy()
        for k in range(1, n):
            p0, p1 = p1, ((2 * k + 1) * t * p1 - k * p0) / (k + 1)
        dp = n * (t * p1 - p0) / (t * t - 1)
        resid = np.abs((p1 + xlo.astype(ld) * dp).astype(np.float64))     # P_n(x + xlo), first order
        plain = np.abs(p1.astype(np.float64))                             # P_n(x)
        assert resid.max() <= 1e-3 * max(plain.max(), 1e-300) + 1e-17 * np.abs(dp.astype(np.float64)).max(), (n, resid.max(), plain.max())


@pytest.mark.parametrize("ab", [(0, 0), (2, 0), (2, 2), (2, -2)])
def test_wigner_tables(oracle, ab):
    import heracles_amd as hx

    lmax = 150
    x = np.array([-0.999, -0.93, -0.2, 0.0, 0.31, 0.9, 0.9985, 0.99995])
    T = hx.wigner_d_table(lmax, ab[0], ab[1], x)
    ref = np.array([oracle.wigner_d(lmax, ab[0], ab[1], float(xx)) for xx in x])
    np.testing.assert_allclose(T, ref, atol=1e-13)


@pytest.mark.parametrize("L,spin", [(16, (0, 0)), (16, (0, 2)), (16, (2, 0)), (40, (0, 0)), (40, (2, 0)), (130, (0, 2))])
def test_mixmat_vs_3j(oracle, L, spin):
    import heracles_amd as hx

    rng = np.random.default_rng(L)
    cl = rng.uniform(0.5, 1.5, L + 1) / (1 + np.arange(L + 1)) ** 2
    out = hx.mixmat(cl, spin=spin)
    ref = oracle.mixmat(cl, spin=spin)
    assert out.shape == (L + 1, L + 1)
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-13 * np.abs(ref).max())


@pytest.mark.parametrize("L", [16, 40, 130])
def test_mixmat_eb_vs_3j(oracle, L):
    import heracles_amd as hx

    rng = np.random.default_rng(L + 1)
    cl = rng.uniform(0.5, 1.5, L + 1) / (1 + np.arange(L + 1)) ** 2
    out = hx.mixmat_eb(cl)
    ref = oracle.mixmat_eb(cl)
    assert out.shape == (3, L + 1, L + 1)
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-13 * np.abs(ref).max())
    np.testing.assert_allclose(out[2], out[0] - out[1], atol=1e-14)


def test_mixmat_shapes_and_identities(oracle):
    import heracles_amd as hx

    L = 48
    cl = np.zeros(L + 1)
    cl[0] = 4 * np.pi
    np.testing.assert_allclose(hx.mixmat(cl), np.eye(L + 1), atol=1e-13)
    eb = hx.mixmat_eb(cl)
    np.testing.assert_allclose(eb[0][2:, 2:], np.eye(L - 1), atol=1e-13)
    np.testing.assert_allclose(eb[1], 0, atol=1e-13)
    rng = np.random.default_rng(2)
    cl = rng.uniform(0.5, 1.5, 31)
    for kw in ({"l1max": 10, "l2max": 20}, {"l1max": 20, "l2max": 10}, {"l1max": 5, "l2max": 5, "l3max": 12},
               {"l1max": 150, "l2max": 129}):
        out = hx.mixmat(cl, spin=(0, 0), **kw)
        ref = oracle.mixmat(cl, spin=(0, 0), **kw)
        assert out.shape == ref.shape
        np.testing.assert_allclose(out, ref, atol=1e-13 * np.abs(ref).max())
        oe, re_ = hx.mixmat_eb(cl, **kw), oracle.mixmat_eb(cl, **kw)
        np.testing.assert_allclose(oe, re_, atol=1e-13 * np.abs(re_).max())
    with pytest.raises(hx.HxError):
        hx.mixmat(cl, spin=(1, 1))


def test_mixmat_large_properties():
    """L = 1024: row-sum rule and E/B relations (oracle-free, size-independent)."""
    import heracles_amd as hx

    L = 1024
    ell = np.arange(L + 1)
    cl = 4 * np.pi * 0.3 * np.exp(-ell * (ell + 1) / 5000.0) + 1e-3 / (1 + ell) ** 2
    M = hx.mixmat(cl, l1max=64, l2max=2 * L + 64, l3max=L)
    target = ((2 * ell + 1) * cl).sum() / (4 * np.pi)
    np.testing.assert_allclose(M.sum(axis=1), target, rtol=1e-10)
    eb = hx.mixmat_eb(cl, l1max=256, l2max=256)
    np.testing.assert_allclose(eb[2], eb[0] - eb[1], atol=1e-12)
    M00 = hx.mixmat(cl, l1max=256, l2max=256)
    # detailed balance: M_{l1 l2} (2 l1 + 1) symmetric
    S = M00 * (2 * np.arange(257) + 1)[:, None]
    np.testing.assert_allclose(S, S.T, atol=1e-11 * np.abs(S).max())


def test_legendre_funcs_golden(golden):
    """heracles.transforms.legendre_funcs (transforms.py:46-112), all three groups incl. d11 / dm11 (:68-73), against the
    values the reference itself produced at 8 nodes (3 of them in its small-angle branch x > 0.998)."""
    from heracles_amd import transforms as tr

    lmax = 40
    for i, x in enumerate(golden["leg/x"]):
        (P, dP), (d11, dm11), (d20, d22, d2m2) = tr.legendre_funcs(lmax, float(x), m=(0, 1, 2))
        np.testing.assert_allclose(P, golden[f"leg/{i}/P"], atol=1e-13)
        np.testing.assert_allclose(dP, golden[f"leg/{i}/dP"], rtol=1e-12, atol=1e-11)
        np.testing.assert_allclose(d11, golden[f"leg/{i}/d11"], atol=1e-12)
        np.testing.assert_allclose(dm11, golden[f"leg/{i}/dm11"], atol=1e-12)
        np.testing.assert_allclose(d20, golden[f"leg/{i}/d20"], atol=1e-12)
        np.testing.assert_allclose(d22, golden[f"leg/{i}/d22"], atol=1e-12)
        np.testing.assert_allclose(d2m2, golden[f"leg/{i}/d2m2"], atol=1e-11)
    assert len(tr.legendre_funcs(lmax, 0.3)) == 2 and len(tr.legendre_funcs(lmax, 0.3, m=(1,))) == 1


@pytest.mark.parametrize("lm", [12, 40, 97])
def test_cl2corr_corr2cl_golden(golden, lm):
    import heracles_amd as hx
    from heracles_amd import transforms as tr

    cls = golden[f"c2c/{lm}/cls"]
    np.testing.assert_allclose(tr._cl2corr(cls), golden[f"c2c/{lm}/corr"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(tr._corr2cl(golden[f"c2c/{lm}/corr"]), golden[f"c2c/{lm}/cls_back"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(tr._cl2corr(cls[:, 0]), golden[f"c2c/{lm}/corr1d"], rtol=1e-9, atol=1e-12)
    # round trip (tests/test_transforms.py:5-13 of the reference)
    back = tr._corr2cl(tr._cl2corr(cls))
    np.testing.assert_allclose(back[2:], cls[2:], rtol=1e-6, atol=1e-9)
    assert hx.gauss_legendre(lm + 1)[0].shape == (lm + 1,)


def test_dict_transforms_and_naturalspice_golden(golden):
    import types

    import heracles_amd as hx
    from heracles_amd.core import Result
    from helpers import key_str

    L = 24
    ell = np.arange(L + 1)
    keys = {("POS", "POS", 0, 0): (0, 0), ("POS", "SHE", 0, 0): (0, 2), ("SHE", "SHE", 0, 0): (2, 2)}
    d = {k: Result(np.array(golden[f"dict/d/{key_str(k)}"]), spin=s, axis=-1, ell=ell) for k, s in keys.items()}
    wd = hx.cl2corr(d)
    back = hx.corr2cl(wd)
    for k in d:
        np.testing.assert_allclose(wd[k].array, golden[f"dict/wd/{key_str(k)}"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(back[k].array, golden[f"dict/back/{key_str(k)}"], rtol=1e-8, atol=1e-12)
    Lm = 2 * L
    ellm = np.arange(Lm + 1)
    fields = {"POS": types.SimpleNamespace(mask="VIS", spin=0), "SHE": types.SimpleNamespace(mask="WHT", spin=2)}
    for tag, tm in (("default", None), ("theta30", 30.0)):
        m = {k: Result(np.array(golden[f"ns/m/{key_str(k)}"]), spin=(0, 0), axis=-1, ell=ellm)
             for k in (("VIS", "VIS", 0, 0), ("VIS", "WHT", 0, 0), ("WHT", "WHT", 0, 0))}
        res = hx.naturalspice(d, m, fields, theta_max=tm)
        for k in d:
            ref = golden[f"ns/{tag}/{key_str(k)}"]
            np.testing.assert_allclose(res[k].array, ref, rtol=1e-6, atol=1e-9 * np.abs(ref).max())


def test_mixing_matrices_batched_context(oracle):
    """The driver through ONE MixmatContext per (l1max, l2max, l3max): every matrix equals the single-call kernels and
    the 3j oracle; hx_mixmat_batch (several masks, one call) gives the same."""
    import ctypes
    import types

    import heracles_amd as hx
    from heracles_amd import _lib, twopoint as tp

    L = 24
    rng = np.random.default_rng(12)
    cls = {("V", "V", 0, 0): rng.uniform(0.1, 1.0, L + 1) / (1.0 + np.arange(L + 1)) ** 2,
           ("V", "W", 0, 1): rng.uniform(0.1, 1.0, L + 1) / (1.0 + np.arange(L + 1)),
           ("W", "W", 1, 1): rng.uniform(0.1, 1.0, L + 1)}
    flds = {"P": types.SimpleNamespace(mask="V", spin=0), "G": types.SimpleNamespace(mask="W", spin=2),
            "K": types.SimpleNamespace(mask="W", spin=0)}
    mms = hx.mixing_matrices(flds, cls, l1max=20, l2max=22, l3max=L)
    assert list(mms) == [t for t, _, _ in tp.mixing_requests(flds, cls)] and len(mms) == 6
    for (f1, f2, i1, i2), res in mms.items():
        spin = (flds[f1].spin, flds[f2].spin)
        cl = cls[flds[f1].mask, flds[f2].mask, i1, i2]
        ref = (oracle.mixmat_eb if all(spin) else oracle.mixmat)(cl, l1max=20, l2max=22, l3max=L, **({} if all(spin) else {"spin": spin}))
        np.testing.assert_allclose(np.asarray(res.array), ref, rtol=1e-11, atol=1e-13 * np.abs(ref).max())
        single = (hx.mixmat_eb if all(spin) else hx.mixmat)(cl, l1max=20, l2max=22, l3max=L, spin=spin)
        np.testing.assert_array_equal(np.asarray(res.array), single)
    # the one-call batch over three masks
    stack = np.ascontiguousarray(np.stack(list(cls.values())))
    n = 3
    o00 = [np.empty((21, 23)) for _ in range(n)]
    o02 = [np.empty((21, 23)) for _ in range(n)]
    oeb = [np.empty((3, 21, 23)) for _ in range(n)]
    kinds = (ctypes.c_int * n)(1 | 2 | 4, 2, 4)
    arr = lambda bufs: (ctypes.c_void_p * n)(*[b.ctypes.data for b in bufs])  # noqa: E731
    _lib.check(_lib.load().hx_mixmat_batch(n, _lib.ptr(stack), L + 1, 20, 22, L, kinds, arr(o00), arr(o02), arr(oeb)))
    for k, cl in enumerate(cls.values()):
        if kinds[k] & 1:
            np.testing.assert_array_equal(o00[k], hx.mixmat(cl, l1max=20, l2max=22, l3max=L, spin=(0, 0)))
        if kinds[k] & 2:
            np.testing.assert_array_equal(o02[k], hx.mixmat(cl, l1max=20, l2max=22, l3max=L, spin=(2, 0)))
        if kinds[k] & 4:
            np.testing.assert_array_equal(oeb[k], hx.mixmat_eb(cl, l1max=20, l2max=22, l3max=L))


@pytest.mark.parametrize("L", [300, 1023, 2047])
def test_mixmat_eb_builds_are_bitwise_repeatable(L):
    """k_mixmat_gemm_dma orders its LDS reads behind loads that write LDS directly by hand (counted vmcnt + barrier): a read that
    came too early would show as a build that differs from the others.  Sizes with one round of tiles, a partly filled round
    and several rounds per XCD; the same screen at L = 4096 / 6144, and against the register-staged kernel, is in
    tools/soak_mixmat.py (profiles/r04_soak_mixmat.txt)."""
    import hashlib

    import heracles_amd as hx

    ell = np.arange(L + 1)
    wl = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / (0.08 * L * L)) + 0.2 / (1.0 + ell) ** 1.5
    seen = {hashlib.sha1(np.ascontiguousarray(hx.mixmat_eb(wl)).tobytes()).hexdigest() for _ in range(6)}
    assert len(seen) == 1


def _mask_spectrum(L):
    ell = np.arange(L + 1)
    # a survey-like mask spectrum: a broad Gaussian core plus a slow power-law tail, so that every l3 up to L contributes
    return 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / 3000.0) + 0.2 / (1.0 + ell) ** 1.5


def _blocks(L):
    """Where the matrices are hard: the high-l corner, the middle of the diagonal, off-diagonal strips at (L, L/2) and
    (L/2, L), and the first spin-2 rows against high columns."""
    h = L // 2
    return [((L - 23, L), (L - 200, L)), ((h - 12, h + 11), (h - 100, h + 100)), ((L - 23, L), (h - 100, h + 100)),
            ((h - 12, h + 11), (L - 200, L)), ((2, 25), (L - 200, L)), ((0, 23), (0, 200))]


@pytest.mark.parametrize("L", [4096, 6144])
def test_mixmat_blocks_at_high_l_vs_3j(oracle, L):
    """BASELINE configs[3] (lmax 4096) and the bench's L = 6144 build against the 3j oracle ON BLOCKS ANYWHERE in the
    matrices (oracle.mixmat_block / mixmat_eb_block: the Schulten-Gordon recursion per (l1, l2), itself pinned at
    these sizes on exact big-integer Racah sums, tests/test_oracle_golden.py): spins (0,0), (2,0), (0,2) and the three
    spin-2 x spin-2 matrices; heracles/twopoint.py:378-388, the roles of the three matrices :445-458."""
    import heracles_amd as hx

    wl = _mask_spectrum(L)
    worst = {}
    for spin in [(0, 0), (2, 0), (0, 2)]:
        got = hx.mixmat(wl, spin=spin)
        assert got.shape == (L + 1, L + 1)
        scale = np.abs(got).max()
        for rows, cols in _blocks(L):
            ref = oracle.mixmat_block(wl, rows, cols, spin=spin)
            blk = got[rows[0]:rows[1] + 1, cols[0]:cols[1] + 1]
            assert np.abs(ref).max() > 0
            err = np.abs(blk - ref).max()
            worst[spin, rows, cols] = err / scale
            assert err <= 1e-12 * scale, (spin, rows, cols, err / scale)
        del got
    eb = hx.mixmat_eb(wl)
    assert eb.shape == (3, L + 1, L + 1)
    scale = np.abs(eb).max()
    for rows, cols in _blocks(L):
        ref = oracle.mixmat_eb_block(wl, rows, cols)
        blk = eb[:, rows[0]:rows[1] + 1, cols[0]:cols[1] + 1]
        for k in range(3):
            err = np.abs(blk[k] - ref[k]).max()
            assert err <= 1e-12 * scale, ("eb", k, rows, cols, err / scale)
    print("worst block error / max|M|:", max(worst.values()))


def test_mixmat_eb_full_sky_identity_lmax4096():
    """A full-sky mask (W_l = 4 pi delta_l0) couples nothing: [0] (EE -> EE) is the identity on l >= 2, [1] (EE -> BB)
    vanishes and [2] = [0] - [1]; so is the mixed (2,0) matrix the identity on l >= 2.  This touches every d^l_{2,+-2} and
    d^l_{2,0} table row and every Gauss-Legendre node of the L = 4096 build (orthogonality of the tables under the nodes)."""
    import heracles_amd as hx

    L = 4096
    one = np.zeros(L + 1)
    one[0] = 4 * np.pi
    eb = hx.mixmat_eb(one)
    ident = np.zeros(L + 1)
    ident[2:] = 1.0
    for k, diag in ((0, ident), (1, 0 * ident), (2, ident)):
        np.testing.assert_allclose(np.diagonal(eb[k]), diag, atol=2e-11)
        off = eb[k].copy()
        np.fill_diagonal(off, 0.0)
        assert np.abs(off).max() <= 2e-11, k
        del off
    del eb
    m20 = hx.mixmat(one, spin=(2, 0))
    np.testing.assert_allclose(np.diagonal(m20), ident, atol=2e-11)
    np.fill_diagonal(m20, 0.0)
    assert np.abs(m20).max() <= 2e-11


def test_mixmat_out_argument_and_pinned_buffers(oracle):
    """`out=`: the caller's array (pageable numpy, page-locked `pinned_empty`, a context's `result_buffer`, a device tensor) receives
    the matrices and is what the call returns; results are bit-identical to the default path whatever the destination."""
    import torch

    import heracles_amd as hx

    L = 300
    cl = _mask_spectrum(L)
    ref = hx.mixmat_eb(cl)
    np.testing.assert_allclose(ref, oracle.mixmat_eb(cl), atol=1e-13 * np.abs(ref).max())
    mine = np.full((3, L + 1, L + 1), np.nan)
    assert hx.mixmat_eb(cl, out=mine) is mine
    np.testing.assert_array_equal(mine, ref)
    pin = hx.pinned_empty((3, L + 1, L + 1))
    pin[:] = np.nan
    assert hx.mixmat_eb(cl, out=pin) is pin
    np.testing.assert_array_equal(pin, ref)
    keep = pin[1, 5:9].copy()
    view = pin[1, 5:9]
    del pin                                     # a live view keeps the page-locked block
    np.testing.assert_array_equal(view, keep)
    dev = torch.empty((3, L + 1, L + 1), dtype=torch.float64, device="cuda")
    assert hx.mixmat_eb(cl, out=dev) is dev
    np.testing.assert_array_equal(dev.cpu().numpy(), ref)
    m00 = hx.mixmat(cl, spin=(0, 0))
    buf = np.empty((L + 1, L + 1))
    assert hx.mixmat(cl, spin=(0, 0), out=buf) is buf
    np.testing.assert_array_equal(buf, m00)
    with hx.MixmatContext(L, L, L) as ctx:
        rb = ctx.result_buffer((2, 2))
        assert rb.shape == (3, L + 1, L + 1) and ctx.result_buffer((2, 2)) is rb
        got = ctx(cl, (2, 2), out=rb)
        assert got is rb
        np.testing.assert_array_equal(rb, ref)
        cl2 = cl * np.linspace(1.0, 0.5, L + 1)
        np.testing.assert_array_equal(ctx(cl2, (2, 2), out=rb), hx.mixmat_eb(cl2))       # the next build overwrites the buffer
        r0 = ctx.result_buffer((0, 0))
        np.testing.assert_array_equal(ctx(cl, (0, 0), out=r0), m00)
        np.testing.assert_array_equal(ctx(cl, (0, 2)), hx.mixmat(cl, spin=(0, 2)))
    with pytest.raises(ValueError):
        hx.mixmat_eb(cl, out=np.empty((3, L + 1, L)))
    # the one-shot calls keep their tables between builds: another size in between, a release, and the results stay what they were
    small = hx.mixmat_eb(cl[:41])
    np.testing.assert_allclose(small, oracle.mixmat_eb(cl[:41]), atol=1e-13 * np.abs(small).max())
    np.testing.assert_array_equal(hx.mixmat_eb(cl), ref)
    from heracles_amd.twopoint import mixmat_release

    mixmat_release()
    np.testing.assert_array_equal(hx.mixmat_eb(cl), ref)
    np.testing.assert_array_equal(hx.mixmat(cl, spin=(0, 0)), m00)
