"""Round-4 widening: ``heracles.io.read_vmap`` (heracles/io.py:360-381) and ``heracles.twopoint.apply_mixing_matrix``
(heracles/twopoint.py:497-524) on the GPU path, against the composition of their parts evaluated by the oracle / by the reference's own
arithmetic in numpy.  healpy files (weights, pixel window) are not available: the window table is synthetic, the files are written with
this package's FITS writer in the layouts healpy produces (float64 / float32 columns, RING / NESTED) -- file-level parity with
healpy-written files is unpinned, as for the other FITS tables."""

import types
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _write_healpix_table(path, columns, nside, ordering="RING", repeat=1024, width=8):
    """A HEALPix map file in the layout healpy.write_map produces: one table, vector columns of `repeat` pixels per row."""
    from heracles_amd import fits as hf

    cols = np.atleast_2d(np.asarray(columns, dtype=np.float64))
    ncols, npix = cols.shape
    rep = min(repeat, npix)
    nrows = npix // rep
    table = np.empty((nrows, ncols, rep), dtype=">f8" if width == 8 else ">f4")
    for c in range(ncols):
        table[:, c, :] = cols[c].reshape(nrows, rep)
    hf._new_file(path, True)
    extra = [hf._card("PIXTYPE", "HEALPIX"), hf._card("ORDERING", ordering), hf._card("NSIDE", nside), hf._card("FIRSTPIX", 0),
             hf._card("LASTPIX", npix - 1), hf._card("INDXSCHM", "IMPLICIT"), hf._card("OBJECT", "FULLSKY")]
    cards = [hf._card("XTENSION", "BINTABLE", "binary table extension"), hf._card("BITPIX", 8), hf._card("NAXIS", 2),
             hf._card("NAXIS1", ncols * rep * width), hf._card("NAXIS2", nrows), hf._card("PCOUNT", 0), hf._card("GCOUNT", 1),
             hf._card("TFIELDS", ncols)]
    for c in range(ncols):
        cards += [hf._card(f"TTYPE{c + 1}", f"COL{c + 1}"), hf._card(f"TFORM{c + 1}", f"{rep}{'D' if width == 8 else 'E'}")]
    cards += extra
    payload = table.tobytes()
    with open(path, "ab") as f:
        f.write(hf._header_bytes(cards))
        f.write(payload)
        f.write(b"\0" * (-len(payload) % hf.BLOCK))


@pytest.mark.parametrize("ordering,width", [("RING", 8), ("NESTED", 4)])
def test_read_vmap_resolution_unseen_and_transform(oracle, tmp_path, ordering, width):
    import heracles_amd as hx
    from heracles_amd.fits import UNSEEN

    nside_file, nside, lmax = 32, 16, 30
    rng = np.random.default_rng(3)
    npix = 12 * nside_file**2
    ring = rng.uniform(0.0, 1.0, npix)
    if width == 4:
        ring = ring.astype(np.float32).astype(np.float64)  # what a float32 file can hold
    ring[rng.choice(npix, 200, replace=False)] = UNSEEN
    other = rng.standard_normal(npix)
    stored = ring if ordering == "RING" else ring[oracle.nest2ring(nside_file, np.arange(npix))]
    stored_other = other if ordering == "RING" else other[oracle.nest2ring(nside_file, np.arange(npix))]
    path = tmp_path / "vmap.fits"
    _write_healpix_table(path, [stored_other, stored], nside_file, ordering=ordering, width=width)
    # as stored, at the file's resolution: unseen pixels are zero, NESTED files come back in RING order
    same = hx.read_vmap(path, field=1)
    want = ring.copy()
    want[want == UNSEEN] = 0.0
    np.testing.assert_array_equal(same, want)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        np.testing.assert_array_equal(hx.read_vmap(path, nside=nside_file, field=1), want)  # same nside: no warning, no regrade
    # another resolution: the reference's warning, hp.ud_grade of the zero-filled map
    with pytest.warns(UserWarning, match="changing NSIDE to 16"):
        low = hx.read_vmap(path, nside=nside, field=1)
    np.testing.assert_array_equal(low, oracle.ud_grade(want, nside))
    # transform: map2alm with (here: unit) weights and three iterations, then almxfl(1 / pw)
    pw = (np.linspace(1.0, 0.8, lmax + 1), np.linspace(1.0, 0.7, lmax + 1))
    with pytest.warns(UserWarning, match="changing NSIDE"):
        alm = hx.read_vmap(path, nside=nside, field=1, transform=True, lmax=lmax, pixwin=pw)
    ref = oracle.map2alm(low[None], nside, lmax, spin=0, niter=3)[0]
    ref = ref * np.concatenate([1.0 / pw[0][m:] for m in range(lmax + 1)])
    np.testing.assert_allclose(alm, ref, atol=1e-11 * np.abs(ref).max())
    with pytest.raises(IndexError):
        hx.read_vmap(path, field=2)


def test_reorder_round_trip_and_against_oracle(oracle):
    import torch

    import heracles_amd as hx

    L = hx._lib.load()
    for nside in (1, 2, 8, 64):
        npix = 12 * nside**2
        m = np.random.default_rng(nside).standard_normal((2, npix))
        nest = np.empty_like(m)
        hx._lib.check(L.hx_reorder(nside, 0, 2, hx._lib.ptr(m), hx._lib.ptr(nest)))
        np.testing.assert_array_equal(nest, m[:, oracle.nest2ring(nside, np.arange(npix))])
        back = torch.empty((2, npix), dtype=torch.float64, device="cuda")
        dn = torch.as_tensor(nest).cuda()
        hx._lib.check(L.hx_reorder(nside, 1, 2, hx._lib.ptr(dn), hx._lib.ptr(back)))
        np.testing.assert_array_equal(back.cpu().numpy(), m)
    with pytest.raises(hx.HxError):
        hx._lib.check(L.hx_reorder(12, 1, 1, hx._lib.ptr(m), hx._lib.ptr(nest)))


def test_apply_mixing_matrix_follows_the_reference(oracle):
    """The arithmetic of heracles/twopoint.py:497-524 restated with numpy on the same inputs (its four cases: scalar x scalar,
    scalar x spin-2, spin-2 x spin-2, rectangular matrices whose output axis sets the angular arrays); host and device matrices."""
    import torch

    import heracles_amd as hx

    rng = np.random.default_rng(17)
    n, m = 37, 53
    ell_out = np.arange(n) + 0.5
    M = {
        ("P", "P", 0, 0): hx.Result(rng.standard_normal((n, m)), spin=(0, 0), axis=-2, ell=ell_out),
        ("P", "G", 0, 1): hx.Result(rng.standard_normal((n, m)), spin=(0, 2), axis=-2),
        ("G", "G", 1, 1): hx.Result(rng.standard_normal((3, n, m)), spin=(2, 2), axis=-2),
    }
    d = {
        ("P", "P", 0, 0): hx.Result(rng.standard_normal(m), spin=(0, 0), axis=-1),
        ("P", "G", 0, 1): hx.Result(rng.standard_normal((2, m)), spin=(0, 2), axis=-1),
        ("G", "G", 1, 1): hx.Result(rng.standard_normal((2, 2, m)).astype(np.float64), spin=(2, 2), axis=-1),
    }
    for device in (False, True):
        Mx = M
        if device:
            Mx = {k: hx.Result(hx.DeviceArray(torch.as_tensor(v.array).cuda(), {}), spin=v.spin, axis=v.axis, ell=v.ell) for k, v in M.items()}
        got = hx.apply_mixing_matrix(d, Mx)
        assert list(got) == list(d)
        k = ("P", "P", 0, 0)
        np.testing.assert_allclose(got[k].array, M[k].array @ d[k].array, rtol=1e-13, atol=1e-13)
        assert got[k].array.shape == (n,)
        np.testing.assert_array_equal(got[k].ell, ell_out)
        k = ("P", "G", 0, 1)
        np.testing.assert_allclose(got[k].array, np.array([M[k].array @ c for c in d[k].array]), rtol=1e-13, atol=1e-13)
        np.testing.assert_array_equal(got[k].ell, np.arange(n))
        np.testing.assert_array_equal(got[k].upper, np.arange(1, n + 1))
        np.testing.assert_array_equal(got[k].weight, np.ones(n))
        k = ("G", "G", 1, 1)
        A, x = M[k].array, d[k].array
        ref = np.array([[A[0] @ x[0, 0] + A[1] @ x[1, 1], A[2] @ x[0, 1]], [A[2] @ x[1, 0], A[1] @ x[0, 0] + A[0] @ x[1, 1]]])
        np.testing.assert_allclose(got[k].array, ref, rtol=1e-13, atol=1e-13)
        assert got[k].spin == (2, 2)


# ---- hx_pinv / invert_mixing_matrix (heracles/twopoint.py:404-494) -------------------------------------------------------------
@pytest.mark.parametrize("shape", [(5, 4), (4, 5), (40, 40), (97, 33), (33, 97), (200, 130), (301, 301)])
def test_pinv_matches_numpy(shape):
    """``hx_pinv`` (blocked one-sided Jacobi SVD) against ``np.linalg.pinv`` with the same ``rcond``: well conditioned, graded singular
    values with a cut in the middle, and exactly rank-deficient input; numpy on the host is the checker here, not a fallback."""
    import heracles_amd as hx
    from heracles_amd.twopoint import pinv

    n, m = shape
    rng = np.random.default_rng(n * 1000 + m)
    a = rng.standard_normal((n, m))
    got, info = pinv(a, 1e-10, info=True)
    ref = np.linalg.pinv(a, rcond=1e-10)
    assert got.shape == (m, n) and info["kept"] == min(n, m)
    np.testing.assert_allclose(got, ref, atol=1e-11 * np.abs(ref).max())
    # graded spectrum 1 ... 1e-8, rcond in a gap: the same singular values are kept
    k = min(n, m)
    u, _, vt = np.linalg.svd(a, full_matrices=False)
    sv = np.logspace(0, -8, k)
    b = (u * sv) @ vt
    for rc in (3e-3, 3e-6):
        got, info = pinv(b, rc, info=True)
        ref = np.linalg.pinv(b, rcond=rc)
        assert info["kept"] == int(np.sum(sv > rc * sv[0]))
        np.testing.assert_allclose(got, ref, atol=1e-9 * np.abs(ref).max())
    # rank one (the reference's own test matrices are all ones: tests/test_twopoint.py:425-447)
    ones = np.ones((n, m))
    got, info = pinv(ones, 1e-4, info=True)
    assert info["kept"] == 1
    np.testing.assert_allclose(got, np.linalg.pinv(ones, rcond=1e-4), atol=1e-13)
    np.testing.assert_allclose(got.sum(), 1.0, rtol=1e-12)
    assert hx is not None


def test_pinv_device_in_out_and_larger_size():
    import torch

    from heracles_amd.twopoint import pinv

    rng = np.random.default_rng(9)
    n = 700
    i, j = np.arange(n)[:, None], np.arange(n)[None, :]
    a = np.exp(-0.5 * ((i - j) / 3.0) ** 2) + 1e-3 * rng.standard_normal((n, n))  # a mixing-matrix-like band
    dev = pinv(torch.as_tensor(a).cuda(), 1e-5, device="cuda")
    assert dev.is_cuda
    ref = np.linalg.pinv(a, rcond=1e-5)
    np.testing.assert_allclose(dev.cpu().numpy(), ref, atol=1e-8 * np.abs(ref).max())


def test_invert_and_apply_mixing_matrix_against_reference_golden():
    """Vectors generated by the reference's own ``invert_mixing_matrix`` / ``apply_mixing_matrix`` (tests/golden/make_golden_mixing.py:
    ``heracles.twopoint`` is importable) on seeded band matrices: square, tall and wide keys of all three spin classes, per-key ``rcond``,
    and matrices of ones."""
    import os

    import heracles_amd as hx

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_mixing.npz"))
    spins = {"POS|POS|0|0": (0, 0), "POS|SHE|0|1": (0, 2), "SHE|SHE|1|1": (2, 2)}
    for name in ("square", "tall", "wide"):
        mats, rconds, cls = {}, {}, {}
        for ks, sp in spins.items():
            key = tuple(int(x) if x.isdigit() else x for x in ks.split("|"))
            M = g[f"{name}/M/{ks}"]
            mats[key] = hx.Result(M, spin=sp, axis=-2, ell=np.arange(M.shape[-2]))
            rconds[key] = float(g[f"{name}/rcond/{ks}"])
            cls[key] = hx.Result(g[f"{name}/cl/{ks}"], spin=sp, axis=-1)
        inv = hx.invert_mixing_matrix(mats, rcond=rconds)
        assert list(inv) == list(mats)
        for ks in spins:
            key = tuple(int(x) if x.isdigit() else x for x in ks.split("|"))
            ref = g[f"{name}/inv/{ks}"]
            assert inv[key].array.shape == ref.shape and inv[key].spin == mats[key].spin
            np.testing.assert_allclose(inv[key].array, ref, atol=1e-9 * np.abs(ref).max())
            np.testing.assert_array_equal(inv[key].ell, g[f"{name}/inv_ell/{ks}"])
        applied = hx.apply_mixing_matrix(cls, inv)
        for ks in spins:
            key = tuple(int(x) if x.isdigit() else x for x in ks.split("|"))
            ref = g[f"{name}/applied/{ks}"]
            np.testing.assert_allclose(applied[key].array, ref, atol=1e-8 * np.abs(ref).max())
        with pytest.raises(KeyError, match="Missing rcond value"):
            hx.invert_mixing_matrix(mats, rcond={})
    ones = {("A", "A", 0, 0): hx.Result(np.ones((11, 21)), spin=(0, 0), axis=-2, ell=np.arange(11)),
            ("B", "B", 0, 0): hx.Result(np.ones((3, 11, 21)), spin=(2, 2), axis=-2, ell=np.arange(11))}
    inv = hx.invert_mixing_matrix(ones, rcond=1e-4)
    np.testing.assert_allclose(inv["A", "A", 0, 0].array, g["ones/inv/A|A|0|0"], atol=1e-13)
    np.testing.assert_allclose(inv["B", "B", 0, 0].array, g["ones/inv/B|B|0|0"], atol=1e-13)
    np.testing.assert_allclose(inv["A", "A", 0, 0].array.sum(), 1.0)   # the reference's assertions (tests/test_twopoint.py:439-447)
    np.testing.assert_allclose(inv["B", "B", 0, 0].array.sum(), 1.5)


def test_pinv_edge_cases():
    """1 x 1, single rows / columns, an all-zero matrix (numpy returns zeros), rcond >= 1 (only the largest singular value survives...
    numpy keeps s > rcond * max(s): none at rcond = 1), bad arguments."""
    import heracles_amd as hx
    from heracles_amd.twopoint import pinv

    rng = np.random.default_rng(0)
    for shape in ((1, 1), (1, 5), (5, 1), (2, 3)):
        a = rng.standard_normal(shape)
        np.testing.assert_allclose(pinv(a, 1e-12), np.linalg.pinv(a, rcond=1e-12), atol=1e-13)
    z = np.zeros((7, 4))
    got, info = pinv(z, 1e-5, info=True)
    np.testing.assert_array_equal(got, np.zeros((4, 7)))
    assert info["kept"] == 0
    a = rng.standard_normal((9, 6))
    np.testing.assert_allclose(pinv(a, 1.0), np.linalg.pinv(a, rcond=1.0), atol=1e-13)   # nothing is strictly above the largest
    np.testing.assert_allclose(pinv(a, 0.5), np.linalg.pinv(a, rcond=0.5), atol=1e-12)
    with pytest.raises(hx.HxError):
        hx._lib.check(hx._lib.load().hx_pinv(3, 3, hx._lib.ptr(np.eye(3)), -1.0, hx._lib.ptr(np.eye(3)), None))


@pytest.mark.parametrize("niter", [0, 3])
def test_read_vmap_transform_equals_mapper_transform_with_a_weight_file(tmp_path, niter):
    """one rule for ``read_vmap(transform=True, datapath=)`` and ``HipHealpixMapper(datapath=).transform`` -- ``niter`` passed
    through unchanged beside the weight file -- so the same map, weights and window give the same alms through both doors."""
    import heracles_amd as hx
    from heracles_amd import weights as hxw

    nside, lmax = 16, 30
    rng = np.random.default_rng(5 + niter)
    npix = 12 * nside**2
    m = rng.uniform(0.0, 1.0, npix)
    hxw.write_compressed_weights(tmp_path / hxw.weights_filename(nside), nside, 1e-2 * rng.standard_normal(hxw.compressed_size(nside)))
    path = tmp_path / "vmap.fits"
    _write_healpix_table(path, [m], nside, ordering="RING", width=8)
    pw = (np.linspace(1.0, 0.8, lmax + 1), np.linspace(1.0, 0.7, lmax + 1))
    alm = hx.read_vmap(path, transform=True, lmax=lmax, pixwin=pw, datapath=tmp_path, niter=niter)
    mapper = hx.HipHealpixMapper(nside, lmax, deconvolve=True, niter=niter, datapath=tmp_path, pixwin=pw)
    np.testing.assert_array_equal(alm, np.asarray(mapper.transform(m, spin=0)))
    # ... and the iteration rule itself is pinned on an independent path: the oracle's weighted analysis with its explicit
    # Jacobi loop -- residual = map - synthesis(alm) against the UNWEIGHTED map, every analysis pass with the pixel weights -- then the
    # window.  (What healpy does for use_pixel_weights=True with iter > 0 stays parity-unpinned: healpy is absent from this image.)
    from oracle import hxoracle as ho

    full = ho.expand_full_weights(nside, hxw.read_compressed_weights(tmp_path / hxw.weights_filename(nside)))
    ref = ho.map2alm(m[None], nside, lmax, spin=0, pix_weights=full, niter=niter)[0]
    fl = 1.0 / pw[0]
    for mm in range(lmax + 1):
        b = mm * (2 * lmax + 1 - mm) // 2
        ref[b + mm : b + lmax + 1] *= fl[mm:]
    assert np.abs(alm - ref).max() <= 1e-11 * np.abs(ref).max()
