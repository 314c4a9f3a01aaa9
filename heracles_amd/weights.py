"""healpy's pixel-weight files for ``HipHealpixMapper`` (``hp.map2alm(..., use_pixel_weights=True, datapath=DATAPATH)``,
heracles/healpy.py:183-189; ``datapath`` from the configuration, heracles/cli.py:536-538).

The files (``healpix_full_weights_nside_NNNN.fits``, one per resolution) are healpy data: they are not part of the reference
tree nor of this repository, so the values themselves are "parity unpinned".  What is implemented -- and tested against a
numpy restatement on synthetic files -- is the wire format (one FITS table column of (nside + 1)(3 nside + 1) / 4 doubles)
and the expansion of that compressed half-quadrant to the full-sky array of multiplicative weights, which runs on the GPU
(``hx_pixel_weights_expand``) and stays in HBM for every later transform.
"""

from __future__ import annotations

import os

import numpy as np

from . import _lib
from . import fits as _fits


def weights_filename(nside: int) -> str:
    return "healpix_full_weights_nside_%04d.fits" % int(nside)


def find_weights_file(datapath, nside):
    """The file for ``nside`` under ``datapath``: healpy keeps it in ``<datapath>/full_weights/``; the directory itself and a
    path to the file are accepted as well.  None if there is none."""
    if datapath is None:
        return None
    datapath = os.fspath(datapath)
    if os.path.isfile(datapath):
        return datapath
    for cand in (os.path.join(datapath, "full_weights", weights_filename(nside)), os.path.join(datapath, weights_filename(nside))):
        if os.path.isfile(cand):
            return cand
    return None


def compressed_size(nside: int) -> int:
    return ((nside + 1) * (3 * nside + 1)) // 4


def read_compressed_weights(path):
    """All values of the first table extension of the file, in file order (float64)."""
    for h, off in _fits._scan(path):
        if str(h.get("XTENSION", "")).strip() != "BINTABLE":
            continue
        cols = _fits._columns(h)
        width = _fits._column_width(h)
        count = h["NAXIS2"] * sum(r for _, r in cols)
        if len(cols) != 1:
            raise NotImplementedError("pixel-weight file with more than one column")
        return np.fromfile(path, dtype=">f8" if width == 8 else ">f4", count=count, offset=off).astype(np.float64)
    raise ValueError(f"{path}: no binary table extension")


def write_compressed_weights(path, nside, compressed):
    """A file in the layout healpy ships (one column of doubles) -- used by the tests and to convert weights computed elsewhere."""
    compressed = np.ascontiguousarray(compressed, dtype=np.float64)
    if compressed.shape != (compressed_size(nside),):
        raise ValueError(f"NSIDE={nside} needs {compressed_size(nside)} compressed weights, got {compressed.shape}")
    _fits._new_file(path, True)
    extra = [_fits._card("NSIDE", int(nside), "Resolution parameter of HEALPIX")]
    _fits._append_table(path, "FULL WEIGHTS", ["COMPRESSED PIXEL WEIGHTS"], 1, compressed.size, compressed.astype(">f8").tobytes(), extra, {})


def expand_pixel_weights(nside, compressed, device=None):
    """Full-sky multiplicative weights ``1 + w`` (12 nside^2) from the compressed values; numpy out, or a CUDA tensor if
    ``device`` is given."""
    _lib.ensure_init()
    compressed = np.ascontiguousarray(compressed, dtype=np.float64)
    npix = 12 * int(nside) ** 2
    if device is None:
        out = np.empty(npix)
    else:
        import torch

        out = torch.empty(npix, dtype=torch.float64, device=device)
    _lib.check(_lib.load().hx_pixel_weights_expand(int(nside), compressed.size, _lib.ptr(compressed), _lib.ptr(out)))
    return out


_cache: dict = {}


def load_pixel_weights(datapath, nside, device="cuda"):
    """Full-sky weights of ``nside`` from ``datapath`` as a device tensor, cached per (file, device); None if there is no file."""
    path = find_weights_file(datapath, nside)
    if path is None:
        return None
    key = (os.path.abspath(path), int(nside), str(device))
    if key not in _cache:
        _cache[key] = expand_pixel_weights(nside, read_compressed_weights(path), device=device)
    return _cache[key]
