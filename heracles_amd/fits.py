"""FITS wire format of maps and alms, either side of the transform path (heracles/io.py:74-218, :569-662).

The reference stores its stage outputs through ``fitsio``: maps as HEALPix tables (RING, IMPLICIT, FULLSKY; one
'D' column per map component, io.py:128-168), alms as the two columns ``real`` / ``imag`` (io.py:189-218), the
dtype-metadata of every array as ``META <KEY>`` header cards (io.py:110-125), the dictionary key as the extension name
(io.py:74-107).  Neither ``fitsio`` nor ``astropy`` is part of this package's requirements, so the container is read and
written here directly from the FITS standard (2880-byte blocks of 80-character cards; BINTABLE extensions; big-endian,
row-major payload), and the payload conversion -- byte swap + (de)interleave of the columns -- runs on the GPU
(``hx_fits_unpack_f64`` / ``hx_fits_pack_f64``), straight into / out of device arrays when asked to
(``device=``), so that `heracles alms` -> `heracles spectra` style stage hand-overs need no host-side array copies.

Files written here carry the cards ``fitsio`` writes for the same call (HIERARCH convention for the ``META`` keys,
``TDIMn`` for vector columns) and are laid out so that the reference's own ``read_maps`` / ``read_alms`` read them;
files written by the reference (any 'D'-column table, vector columns included) are read.  Parity with fitsio-written
files is pinned on the FITS standard only: the reference tree holds no FITS fixture and fitsio is absent here.
"""

from __future__ import annotations

import os
import re
from collections.abc import MutableMapping, Sequence

import numpy as np

from . import _lib
from .core import toc_match

BLOCK = 2880
_COMMENTS = {
    "catalog": "catalog of map", "spin": "spin weight of map", "kernel": "mapping kernel of map",
    "nside": "NSIDE parameter of HEALPix map", "lmax": "LMAX parameter of map", "deconv": "pixel window function was deconvolved",
    "bias": "additive bias of spectrum",
}


# ---- keys <-> extension names (heracles/io.py:74-107) -----------------------------------------------------------------
def string_from_key(key):
    """Extension name of a dictionary key: parts joined by '-', literal backslashes and dashes escaped, anything
    outside printable ASCII replaced by '~'."""
    if isinstance(key, Sequence) and not isinstance(key, str):
        return "-".join(string_from_key(part) for part in key)
    text = str(key).replace("\\", "\\\\").replace("-", "\\-")
    return re.sub(r"[^ -~]+", "~", text, flags=re.ASCII)


def key_from_string(text):
    """Inverse of :func:`string_from_key`; parts that are (signed) digit strings become ints."""
    pieces = re.split(r"(?<!\\)-", text.replace("\\\\", "\0"))
    if len(pieces) > 1:
        return tuple(key_from_string(p) for p in pieces)
    part = pieces[0].replace("\\-", "-").replace("\0", "\\")
    digits = part[1:] if part.startswith("-") else part
    return int(part) if digits.isdigit() else part


# ---- header cards --------------------------------------------------------------------------------------------------
def _format_value(value):
    if isinstance(value, (bool, np.bool_)):
        return "T" if value else "F"
    if isinstance(value, (int, np.integer)):
        return str(int(value))
    if isinstance(value, (float, np.floating)):
        text = repr(float(value)).upper()
        return text if ("." in text or "E" in text or "N" in text) else text + "."
    text = str(value).replace("'", "''")
    return "'" + text.ljust(8) + "'"


def _card(key, value=None, comment=""):
    key = key.upper()
    if value is None:
        return key.ljust(80)[:80]
    val = _format_value(value)
    if len(key) <= 8 and " " not in key:
        body = f"{key:<8}= {val:>20}" if not val.startswith("'") else f"{key:<8}= {val:<20}"
    else:  # ESO HIERARCH convention, what fitsio writes for long keywords and keywords with spaces
        body = f"HIERARCH {key} = {val}"
    if comment:
        body += " / " + comment
    return body.ljust(80)[:80]


def _header_bytes(cards):
    text = "".join(cards) + "END".ljust(80)
    text += " " * (-len(text) % BLOCK)
    return text.encode("ascii")


def _parse_value(text):
    text = text.strip()
    if text.startswith("'"):
        end = 1
        while True:  # closing quote, '' is an escaped quote
            end = text.index("'", end)
            if text[end : end + 2] == "''":
                end += 2
                continue
            break
        return text[1:end].replace("''", "'").rstrip()
    text = text.split("/", 1)[0].strip()
    if text in ("T", "F"):
        return text == "T"
    try:
        return int(text)
    except ValueError:
        pass
    try:
        return float(text.replace("D", "E"))
    except ValueError:
        return text


def _read_header(f):
    """Header of the HDU at the file position: ({keyword: value}, ordered) -- or None at end of file."""
    cards = {}
    while True:
        block = f.read(BLOCK)
        if not block:
            return None if not cards else cards
        if len(block) < BLOCK:
            raise ValueError("truncated FITS header")
        for i in range(0, BLOCK, 80):
            card = block[i : i + 80].decode("ascii", errors="replace")
            if card.startswith("END") and card[3:].strip() == "":
                return cards
            if card.startswith("HIERARCH "):
                left, _, right = card[9:].partition("=")
                cards[left.strip()] = _parse_value(right)
            elif card[8:10] == "= ":
                cards[card[:8].strip()] = _parse_value(card[10:])


def _data_bytes(h):
    if h.get("NAXIS", 0) == 0:
        return 0
    n = abs(h.get("BITPIX", 8)) // 8
    for i in range(1, h["NAXIS"] + 1):
        n *= h[f"NAXIS{i}"]
    return h.get("GCOUNT", 1) * (n + h.get("PCOUNT", 0))


def _scan(path):
    """[(header, payload offset)] of every HDU of the file."""
    out = []
    with open(path, "rb") as f:
        while True:
            h = _read_header(f)
            if h is None:
                break
            off = f.tell()
            out.append((h, off))
            size = _data_bytes(h)
            f.seek(off + size + (-size % BLOCK))
    return out


def _metadata(h):
    return {k[5:].lower(): v for k, v in h.items() if k.startswith("META ")}


def _meta_cards(md):
    return [_card("META " + k.upper(), v, _COMMENTS.get(k, "")) for k, v in (md or {}).items()]


def _columns(h):
    """[(name, repeat)] of a table whose columns are all floats of one width: 'rD' (8 bytes) or 'rE' (4 bytes, what healpy's
    write_map produces for masks and visibility maps by default).  The width is returned by _column_width."""
    cols = []
    for i in range(1, h["TFIELDS"] + 1):
        m = re.fullmatch(r"(\d*)([DE])", str(h[f"TFORM{i}"]).strip())
        if not m:
            raise NotImplementedError(f"column {i} has TFORM {h[f'TFORM{i}']!r}: only float64 ('D') and float32 ('E') columns are supported")
        cols.append((str(h.get(f"TTYPE{i}", f"COL{i}")).strip(), int(m.group(1) or 1)))
    return cols


def _column_width(h):
    kinds = {re.fullmatch(r"(\d*)([DE])", str(h[f"TFORM{i}"]).strip()).group(2) for i in range(1, h["TFIELDS"] + 1)}
    if len(kinds) != 1:
        raise NotImplementedError("columns of mixed float widths")
    return 8 if kinds == {"D"} else 4


def _tdim(h, i):
    """Leading dimensions (C order) of vector column i from its TDIMi card '(d_fast,...,d_slow)', or None."""
    t = h.get(f"TDIM{i}")
    if t is None:
        return None
    dims = [int(x) for x in str(t).strip().strip("()").split(",") if x.strip()]
    return tuple(reversed(dims))


def _new_file(path, clobber):
    if not os.path.isfile(path) or clobber:
        with open(path, "wb") as f:
            f.write(_header_bytes([_card("SIMPLE", True, "file does conform to FITS standard"), _card("BITPIX", 16, "number of bits per data pixel"),
                                   _card("NAXIS", 0, "number of data axes"), _card("EXTEND", True, "FITS dataset may contain extensions")]))


def _append_table(path, ext, names, repeat, nrows, payload, extra_cards, md, lead=None):
    """One BINTABLE extension of len(names) columns 'repeat D' with the given big-endian row-major payload; lead: the
    leading dimensions (C order) a row's vector stands for -- written as TDIM in FITS (fastest-first) order, as fitsio does
    for the moveaxis layout of heracles/io.py:189-200."""
    nb = 8 * repeat * len(names)
    cards = [_card("XTENSION", "BINTABLE", "binary table extension"), _card("BITPIX", 8, "8-bit bytes"), _card("NAXIS", 2, "2-dimensional binary table"),
             _card("NAXIS1", nb, "width of table in bytes"), _card("NAXIS2", nrows, "number of rows in table"),
             _card("PCOUNT", 0, "size of special data area"), _card("GCOUNT", 1, "one data group (required keyword)"),
             _card("TFIELDS", len(names), "number of fields in each row")]
    for i, name in enumerate(names, start=1):
        cards.append(_card(f"TTYPE{i}", name, f"label for field {i:3d}"))
        cards.append(_card(f"TFORM{i}", "D" if repeat == 1 else f"{repeat}D", "data format of field: 8-byte DOUBLE"))
        if repeat > 1:
            dims = tuple(lead) if lead else (repeat,)
            cards.append(_card(f"TDIM{i}", "(" + ",".join(str(d) for d in reversed(dims)) + ")", "dimensions of field"))
    cards.append(_card("EXTNAME", ext, "name of this binary table extension"))
    cards += extra_cards + _meta_cards(md)
    with open(path, "ab") as f:
        f.write(_header_bytes(cards))
        nbytes = memoryview(payload).nbytes
        f.write(payload)
        f.write(b"\0" * (-nbytes % BLOCK))


def _to_table(array, nrows, nc1, nc2, s1, s2, srow):
    """Big-endian row-major payload (bytes-like) of a native array (numpy or device tensor) through the GPU."""
    table = np.empty(nrows * nc1 * nc2, dtype=np.float64)
    _lib.ensure_init()
    _lib.check(_lib.load().hx_fits_pack_f64(nrows, nc1, nc2, s1, s2, srow, _lib.ptr(array), _lib.ptr(table)))
    return table


def _from_table(path, off, nrows, nc1, nc2, s1, s2, srow, out, width=8):
    if width == 4:
        # float32 columns: widened on the host, then the same device pass as float64 tables (the kernel swaps 8-byte words)
        raw = np.fromfile(path, dtype=">f4", count=nrows * nc1 * nc2, offset=off).astype(">f8").view(np.float64)
    else:
        raw = np.fromfile(path, dtype=np.float64, count=nrows * nc1 * nc2, offset=off)  # bytes as they are in the file
    _lib.ensure_init()
    _lib.check(_lib.load().hx_fits_unpack_f64(nrows, nc1, nc2, s1, s2, srow, _lib.ptr(raw), _lib.ptr(out)))
    return out


def _empty(shape, complex_, device):
    if device is None:
        return np.empty(shape, dtype=np.complex128 if complex_ else np.float64)
    import torch

    return torch.empty(shape, dtype=torch.complex128 if complex_ else torch.float64, device=device)


def _with_metadata(arr, md, device):
    if device is None:
        arr.dtype = np.dtype(arr.dtype, metadata=md)
        return arr
    from .core import DeviceArray

    return DeviceArray(arr, md)


def _host_or_tensor(a, dtype):
    if hasattr(a, "tensor"):  # DeviceArray
        return a.tensor.contiguous(), dict(a.dtype.metadata or {})
    if hasattr(a, "data_ptr"):
        return a.contiguous(), {}
    return np.ascontiguousarray(a, dtype=dtype), dict(a.dtype.metadata or {}) if hasattr(a, "dtype") else {}


# ---- maps (heracles/io.py:128-187, :383-440) ------------------------------------------------------------------------------
def _write_map(path, ext, m):
    arr, md = _host_or_tensor(m, np.float64)
    npix = arr.shape[-1]
    ncols = 1
    for d in arr.shape[:-1]:
        ncols *= d
    nside = int(round((npix / 12) ** 0.5))
    if 12 * nside * nside != npix:
        raise ValueError("Wrong pixel number (it is not 12*nside**2)")
    names = ["MAP"] if ncols == 1 else [f"MAP{j}" for j in range(1, ncols + 1)]
    payload = _to_table(arr, npix, ncols, 1, npix, 0, 1)
    extra = [_card("PIXTYPE", "HEALPIX", "HEALPIX pixelisation"), _card("ORDERING", "RING", "Pixel ordering scheme, either RING or NESTED"),
             _card("NSIDE", nside, "Resolution parameter of HEALPIX"), _card("FIRSTPIX", 0, "First pixel # (0 based)"),
             _card("LASTPIX", npix - 1, "Last pixel # (0 based)"), _card("INDXSCHM", "IMPLICIT", "Indexing: IMPLICIT or EXPLICIT"),
             _card("OBJECT", "FULLSKY", "Sky coverage, either FULLSKY or PARTIAL")]
    _append_table(path, ext, names, 1, npix, payload, extra, md)


def _read_map(path, h, off, device=None):
    cols = _columns(h)
    rep = cols[0][1]
    if any(r != rep for _, r in cols):
        raise NotImplementedError("columns of different repeat counts")
    nrows, ncols = h["NAXIS2"], len(cols)
    npix = nrows * rep
    out = _empty((ncols, npix) if ncols > 1 else (npix,), False, device)
    _from_table(path, off, nrows, ncols, rep, npix, 1, rep, out, _column_width(h))
    return _with_metadata(out, _metadata(h), device)


# ---- complex arrays (heracles/io.py:189-218) ---------------------------------------------------------------------------------
def _write_complex(path, ext, a):
    arr, md = _host_or_tensor(a, np.complex128)
    n = arr.shape[-1]
    rep = 1
    for d in arr.shape[:-1]:
        rep *= d
    payload = _to_table(arr, n, 2, rep, 1, 2 * n, 2)
    _append_table(path, ext, ["real", "imag"], rep, n, payload, [], md, lead=tuple(arr.shape[:-1]))


def _read_complex(path, h, off, device=None):
    cols = dict(_columns(h))
    if "real" not in cols or "imag" not in cols or list(cols)[:2] != ["real", "imag"] or cols["real"] != cols["imag"]:
        raise NotImplementedError("expected the columns 'real', 'imag' of equal shape")
    rep, n = cols["real"], h["NAXIS2"]
    if _column_width(h) != 8:
        raise NotImplementedError("complex arrays are stored as float64 columns (heracles/io.py:189-200)")
    out = _empty((rep, n) if rep > 1 else (n,), True, device)
    _from_table(path, off, n, 2, rep, 1, 2 * n, 2, out)
    lead = _tdim(h, 1)
    if lead is not None and len(lead) > 1 and int(np.prod(lead)) == rep:
        out = out.reshape(*lead, n)  # the moveaxis(0, -1) layout of io.py:203-218
    return _with_metadata(out, _metadata(h), device)


# ---- dictionaries of maps / alms ------------------------------------------------------------------------------------------
def _write_all(path, items, writer, clobber):
    _new_file(path, clobber)
    for key, value in items.items():
        writer(path, string_from_key(key), value)


def _read_all(path, reader, include, exclude, device):
    out = {}
    for h, off in _scan(path):
        ext = str(h.get("EXTNAME", "")).strip()
        if h.get("XTENSION") != "BINTABLE" or not ext:
            continue
        key = key_from_string(ext)
        if not key or not toc_match(key, include, exclude):
            continue
        out[key] = reader(path, h, off, device)
    return out


def write_maps(path, maps, *, clobber=False):
    """Write a set of maps (numpy arrays or device tensors) to a FITS file; appends unless ``clobber`` (io.py:383-410)."""
    _write_all(path, maps, _write_map, clobber)


def read_maps(path, *, include=None, exclude=None, device=None):
    """Read a set of maps; ``device="cuda"`` returns DeviceArrays whose payload was decoded straight into HBM."""
    return _read_all(path, _read_map, include, exclude, device)


def write_alms(path, alms, *, clobber=False):
    """Write a set of alms (numpy arrays, device tensors or DeviceArrays) to a FITS file (io.py:443-470)."""
    _write_all(path, alms, _write_complex, clobber)


def read_alms(path, *, include=None, exclude=None, device=None):
    return _read_all(path, _read_complex, include, exclude, device)


class _FitsDict(MutableMapping):
    """A FITS-backed mapping (heracles/io.py:569-650): items are written as they are set, read when asked for."""

    _reader = _writer = None

    def __init__(self, path, *, clobber=False, device=None):
        self.path, self.device = os.fspath(path), device
        _new_file(self.path, clobber)

    def _tables(self):
        for h, off in _scan(self.path):
            ext = str(h.get("EXTNAME", "")).strip()
            if h.get("XTENSION") == "BINTABLE" and ext and key_from_string(ext):
                yield ext, h, off

    def __iter__(self):
        for ext, _, _ in self._tables():
            yield key_from_string(ext)

    def __len__(self):
        return sum(1 for _ in self._tables())

    def __contains__(self, key):
        want = string_from_key(key)
        return any(ext == want for ext, _, _ in self._tables())

    def __getitem__(self, key):
        want = string_from_key(key)
        for ext, h, off in self._tables():
            if ext == want:
                return type(self)._reader(self.path, h, off, self.device)
        raise KeyError(want)

    def __setitem__(self, key, value):
        type(self)._writer(self.path, string_from_key(key), value)

    def __delitem__(self, key):
        raise NotImplementedError("deleting FITS extensions is not supported")


class MapFits(_FitsDict):
    _reader, _writer = staticmethod(_read_map), staticmethod(_write_map)


class AlmFits(_FitsDict):
    _reader, _writer = staticmethod(_read_complex), staticmethod(_write_complex)


# ---- visibility maps (heracles/io.py:360-381) ------------------------------------------------------------------------------------
UNSEEN = -1.6375e30  # healpy.UNSEEN


def read_vmap(filename, nside=None, field=0, *, transform=False, lmax=None, pixwin=None, datapath=None, niter=3):
    """``heracles.io.read_vmap`` (heracles/io.py:360-381): column ``field`` of the first HEALPix table of the file as a RING map
    (a NESTED file is reordered, as ``hp.read_map`` does), unseen pixels set to zero, changed to ``nside`` with a warning if the file has
    another resolution (``hx_ud_grade``); ``transform=True``: its alms up to ``lmax`` (healpy's default: 3 nside - 1) with pixel weights,
    divided by the pixel window (``hx_map2alm`` with ``fl = 1 / pw``).

    Not in the reference's signature, because healpy's data files are not available here: ``pixwin=(pw_T, pw_P)`` (the window table; else
    healpy's, if installed), ``datapath`` (directory of healpy's weight files; without one the quadrature uses unit weights), ``niter``
    (Jacobi iterations -- healpy's ``map2alm`` default of three, which the reference's call does not override: heracles/io.py:377).
    ONE rule for this function and ``HipHealpixMapper.transform``: ``niter`` is passed through unchanged whether or not a weight file
    is used, so the two give the same alms for the same map, weights and window (tests/test_gpu_widen.py; which of weights + 0 or
    weights + 3 iterations healpy itself runs is parity-unpinned: healpy is not importable here)."""
    from warnings import warn

    from .mapper import pixel_window, ud_grade

    hdus = [(h, off) for h, off in _scan(filename) if str(h.get("XTENSION", "")).strip() == "BINTABLE"]
    if not hdus:
        raise ValueError(f"{filename}: no binary table extension")
    h, off = hdus[0]
    cols = _read_map(filename, h, off)
    cols = np.atleast_2d(np.asarray(cols))
    if not 0 <= field < cols.shape[0]:
        raise IndexError(f"{filename}: field {field} of {cols.shape[0]}")
    vmap = np.array(cols[field], dtype=np.float64)
    npix = vmap.shape[0]
    nside_in = int(round((npix / 12) ** 0.5))
    if 12 * nside_in * nside_in != npix:
        raise ValueError(f"{filename}: {npix} pixels are not a full-sky HEALPix map")
    if str(h.get("ORDERING", "RING")).strip().upper().startswith("NEST"):
        ring = np.empty_like(vmap)
        _lib.ensure_init()
        _lib.check(_lib.load().hx_reorder(nside_in, 1, 1, _lib.ptr(vmap), _lib.ptr(ring)))
        vmap = ring
    # set unseen pixels to zero.  hp.read_map has already replaced everything its mask_bad accepts (|v - UNSEEN| <= 1e-15 + 1e-5 |UNSEEN|:
    # the float32 image of UNSEEN in the files healpy writes by default) with the exact UNSEEN the reference compares to -- healpy's
    # source is not available here: restated from its published behaviour, unpinned
    vmap[np.abs(vmap - UNSEEN) <= 1e-15 + 1e-5 * abs(UNSEEN)] = 0.0
    if nside is not None and nside != nside_in:
        warn(f"{filename}: changing NSIDE to {nside}")  # vmap is provided at a different resolution
        vmap = ud_grade(vmap, nside)
    if transform:
        from . import sht
        from .weights import load_pixel_weights

        nside_t = int(round((vmap.shape[0] / 12) ** 0.5))
        lmax_t = 3 * nside_t - 1 if lmax is None else int(lmax)
        pw = pixel_window(nside_t, lmax_t, pixwin)[0]
        weights = load_pixel_weights(datapath, nside_t) if datapath is not None else None
        if datapath is not None and weights is None:
            raise FileNotFoundError(f"no pixel-weight file for NSIDE={nside_t} under datapath {datapath!r}")
        plan = sht.get_plan(nside_t, lmax_t)
        vmap = plan.map2alm(vmap[None], 0, pix_weights=weights, fl=1.0 / pw, niter=niter)[0]
    return vmap
