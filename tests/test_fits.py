"""FITS wire format of maps and alms (heracles/io.py:74-218).  fitsio / astropy are absent from this image and the
reference tree holds no FITS fixture, so file-level parity is pinned on the FITS standard: the container written here is
checked byte by byte against an independent numpy construction of the same table (big-endian, row-major), and the key
encoding against the cases of the reference's own tests (tests/test_io.py: keys with dashes, backslashes, ints, tuples)."""

import numpy as np
import pytest


def test_key_encoding_round_trip():
    from heracles_amd.fits import key_from_string, string_from_key

    cases = {("POS", 1): "POS-1", ("a-b", "c\\d", 2): "a\\-b-c\\\\d-2", "plain": "plain", 7: "7", ("X", -3): "X-\\-3",
             ("P", "G_E", 0, 1): "P-G_E-0-1", ("é", 1): "~-1"}
    for key, text in cases.items():
        assert string_from_key(key) == text
        if "~" not in text:
            assert key_from_string(text) == key
    assert key_from_string("A-B-1-2") == ("A", "B", 1, 2)
    assert key_from_string("12") == 12 and key_from_string("x\\-y") == "x-y"


def test_cards_and_header_scan(tmp_path):
    from heracles_amd import fits as hf

    assert hf._card("NAXIS", 2, "c") == "NAXIS   =                    2 / c".ljust(80)
    assert hf._card("EXTNAME", "POS-1").startswith("EXTNAME = 'POS-1   '")
    assert hf._card("META SPIN", 2, "spin weight of map").startswith("HIERARCH META SPIN = 2 / spin weight of map")
    assert hf._card("META DECONV", True).startswith("HIERARCH META DECONV = T")
    assert len(hf._header_bytes([hf._card("SIMPLE", True)])) == 2880
    # a hand-assembled table header as fitsio writes it is parsed, metadata included
    cards = [hf._card("XTENSION", "BINTABLE"), hf._card("BITPIX", 8), hf._card("NAXIS", 2), hf._card("NAXIS1", 16), hf._card("NAXIS2", 3),
             hf._card("PCOUNT", 0), hf._card("GCOUNT", 1), hf._card("TFIELDS", 2), hf._card("TTYPE1", "real"), hf._card("TFORM1", "D"),
             hf._card("TTYPE2", "imag"), hf._card("TFORM2", "D"), hf._card("EXTNAME", "SHE-2"),
             "HIERARCH META NSIDE = 64 / NSIDE parameter of HEALPix map".ljust(80), "HIERARCH META KERNEL = 'healpix' / mapping kernel".ljust(80),
             "HIERARCH META BIAS = 1.25E-07".ljust(80), "COMMENT free text = not a value".ljust(80)]
    p = tmp_path / "t.fits"
    with open(p, "wb") as f:
        f.write(hf._header_bytes([hf._card("SIMPLE", True), hf._card("BITPIX", 16), hf._card("NAXIS", 0), hf._card("EXTEND", True)]))
        f.write(hf._header_bytes(cards))
        f.write(np.arange(6, dtype=">f8").tobytes() + b"\0" * (2880 - 48))
    hdus = hf._scan(p)
    assert len(hdus) == 2 and hdus[1][1] == 2 * 2880
    h = hdus[1][0]
    assert hf._columns(h) == [("real", 1), ("imag", 1)]
    assert hf._metadata(h) == {"nside": 64, "kernel": "healpix", "bias": 1.25e-07}
    assert hf.key_from_string(h["EXTNAME"]) == ("SHE", 2)


@pytest.mark.gpu
def test_maps_and_alms_round_trip_and_bytes(tmp_path):
    import torch

    import heracles_amd as hx
    from heracles_amd import fits as hf

    rng = np.random.default_rng(5)
    nside, lmax = 8, 12
    npix, nlm = 12 * nside**2, (lmax + 1) * (lmax + 2) // 2
    pos = rng.standard_normal(npix)
    hx.update_metadata(pos, spin=0, nside=nside, kernel="healpix", catalog="cat-1.fits", deconv=True)
    she = rng.standard_normal((2, npix))
    hx.update_metadata(she, spin=2, nside=nside, bias=1.5e-7)
    maps = {("POS", 1): pos, ("SHE", 1): she}
    p = tmp_path / "maps.fits"
    hf.write_maps(p, maps, clobber=True)
    back = hf.read_maps(p)
    assert list(back) == list(maps)
    for k in maps:
        np.testing.assert_array_equal(back[k], maps[k])
        assert back[k].dtype.metadata == maps[k].dtype.metadata
    # payload bytes = what the FITS standard prescribes: rows of big-endian doubles, one value per column
    hdus = hf._scan(p)
    h, off = hdus[2]
    assert (h["TFIELDS"], h["NAXIS1"], h["NAXIS2"], h["TTYPE1"], h["TTYPE2"], h["ORDERING"], h["NSIDE"], h["LASTPIX"]) == \
        (2, 16, npix, "MAP1", "MAP2", "RING", nside, npix - 1)
    raw = np.fromfile(p, dtype=">f8", count=2 * npix, offset=off).reshape(npix, 2)
    np.testing.assert_array_equal(raw, she.T)
    assert (p.stat().st_size % 2880) == 0
    # device in / device out
    dmaps = hf.read_maps(p, device="cuda", include=[("SHE",)])
    assert list(dmaps) == [("SHE", 1)] and dmaps["SHE", 1].tensor.is_cuda
    np.testing.assert_array_equal(dmaps["SHE", 1].tensor.cpu().numpy(), she)
    assert dmaps["SHE", 1].dtype.metadata["bias"] == 1.5e-7
    # alms: 1-d and (2, nlm), numpy and DeviceArray sources, append mode
    a0 = rng.standard_normal(nlm) + 1j * rng.standard_normal(nlm)
    hx.update_metadata(a0, spin=0, nside=nside)
    a2 = rng.standard_normal((2, nlm)) + 1j * rng.standard_normal((2, nlm))
    hx.update_metadata(a2, spin=2, nside=nside, deconv=False)
    q = tmp_path / "alms.fits"
    hf.write_alms(q, {("POS", 1): a0}, clobber=True)
    hf.write_alms(q, {("SHE", 1): hx.DeviceArray(torch.as_tensor(np.asarray(a2)).cuda(), a2.dtype.metadata)})
    alms = hf.read_alms(q)
    np.testing.assert_array_equal(alms["POS", 1], a0)
    np.testing.assert_array_equal(alms["SHE", 1], a2)
    assert alms["SHE", 1].dtype.metadata == a2.dtype.metadata and alms["SHE", 1].shape == (2, nlm)
    h, off = hf._scan(q)[2]
    assert (h["TFORM1"], h["TFORM2"], h["TTYPE1"], h["TTYPE2"], h["TDIM1"]) == ("2D", "2D", "real", "imag", "(2)")
    raw = np.fromfile(q, dtype=">f8", count=4 * nlm, offset=off).reshape(nlm, 2, 2)  # (row, real|imag, component)
    np.testing.assert_array_equal(raw[:, 0, :], a2.real.T)
    np.testing.assert_array_equal(raw[:, 1, :], a2.imag.T)
    # the FITS-backed mappings: set, iterate, get, straight into HBM
    d = hf.AlmFits(tmp_path / "d.fits", clobber=True, device="cuda")
    d["SHE", 2] = a2
    d["POS", 2] = a0
    assert list(d) == [("SHE", 2), ("POS", 2)] and ("POS", 2) in d and ("X", 1) not in d and len(d) == 2
    np.testing.assert_array_equal(d["SHE", 2].tensor.cpu().numpy(), a2)
    # a HEALPix file with 1024-element vector columns (the layout healpy.write_map produces) is read as well
    big = rng.standard_normal((2, 12 * 32**2))
    v = tmp_path / "vec.fits"
    hf._new_file(v, True)
    table = np.ascontiguousarray(big.reshape(2, -1, 1024).transpose(1, 0, 2)).astype(">f8").tobytes()
    hf._append_table(v, "VMAP", ["T", "Q"], 1024, big.shape[1] // 1024, np.frombuffer(table, dtype=np.uint8), [], {})
    np.testing.assert_array_equal(hf.read_maps(v)["VMAP"], big)
    # float32 vector columns ('1024E': healpy.write_map's default for masks / visibility maps) are widened to float64
    v32 = tmp_path / "vec32.fits"
    hf._new_file(v32, True)
    big32 = big.astype(np.float32)
    table32 = np.ascontiguousarray(big32.reshape(2, -1, 1024).transpose(1, 0, 2)).astype(">f4").tobytes()
    cards = [hf._card("XTENSION", "BINTABLE"), hf._card("BITPIX", 8), hf._card("NAXIS", 2), hf._card("NAXIS1", 2 * 4 * 1024),
             hf._card("NAXIS2", big.shape[1] // 1024), hf._card("PCOUNT", 0), hf._card("GCOUNT", 1), hf._card("TFIELDS", 2),
             hf._card("TTYPE1", "T"), hf._card("TFORM1", "1024E"), hf._card("TTYPE2", "Q"), hf._card("TFORM2", "1024E"), hf._card("EXTNAME", "VMAP")]
    with open(v32, "ab") as f:
        f.write(hf._header_bytes(cards))
        f.write(table32)
        f.write(b"\0" * (-len(table32) % 2880))
    np.testing.assert_array_equal(hf.read_maps(v32)["VMAP"], big32.astype(np.float64))
    # alms with two leading axes keep their shape: TDIM carries the reversed leading dimensions of the reference's
    # moveaxis layout (heracles/io.py:189-218)
    a3 = rng.standard_normal((3, 2, nlm)) + 1j * rng.standard_normal((3, 2, nlm))
    hx.update_metadata(a3, spin=2, nside=nside)
    r = tmp_path / "alms3.fits"
    hf.write_alms(r, {("SHE", 7): a3}, clobber=True)
    h3, off3 = hf._scan(r)[1]
    assert (h3["TFORM1"], h3["TDIM1"], h3["TDIM2"]) == ("6D", "(2,3)", "(2,3)")
    raw3 = np.fromfile(r, dtype=">f8", count=12 * nlm, offset=off3).reshape(nlm, 2, 3, 2)  # (row, real|imag, a, b)
    np.testing.assert_array_equal(raw3[:, 0], np.moveaxis(a3.real, -1, 0))
    back3 = hf.read_alms(r)["SHE", 7]
    assert back3.shape == (3, 2, nlm)
    np.testing.assert_array_equal(back3, a3)
