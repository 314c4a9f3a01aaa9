"""HEALPix mapper backed by the HIP spherical harmonic transform.

``HipHealpixMapper`` satisfies the ``heracles.mapper.Mapper`` protocol
(heracles/mapper.py:33-74) and mirrors ``heracles.healpy.HealpixMapper``
(heracles/healpy.py:68-209): same constructor, properties, dtype-metadata and spin
dispatch; ``transform`` runs on the GPU through libhxsht instead of ``healpy.map2alm``.

Data healpy ships but this repository cannot (pixel-window tables, pixel-weight files)
is taken from the caller: pass ``pixwin=(pw_T, pw_P)`` / ``pixel_weights=...``; if they
are omitted and the user's environment has healpy, its tables are used.
"""

from __future__ import annotations

import numpy as np

from . import sht
from .core import DeviceArray, update_metadata


def pixel_window(nside, lmax, pixwin=None):
    """(pw_T, pw_P) up to lmax: explicit arrays, a callable(nside, lmax), or healpy's table."""
    if pixwin is not None:
        if callable(pixwin):
            pw0, pw2 = pixwin(nside, lmax)
        else:
            pw0, pw2 = pixwin
        pw0, pw2 = np.asarray(pw0, dtype=float), np.asarray(pw2, dtype=float)
        if pw0.shape[-1] < lmax + 1 or pw2.shape[-1] < lmax + 1:
            raise ValueError("pixel window shorter than lmax+1")
        return pw0[: lmax + 1], pw2[: lmax + 1]
    try:
        import healpy as hp
    except ImportError as exc:
        raise RuntimeError(
            "pixel-window deconvolution needs the HEALPix pixel window table: pass "
            "pixwin=(pw_T, pw_P) to HipHealpixMapper (or install healpy for its table), "
            "or use deconvolve=False"
        ) from exc
    return hp.pixwin(nside, lmax=lmax, pol=True)


def _check_points(rc):
    """healpy.ang2pix raises ValueError for points outside 0 <= theta <= pi (or NaN); libhxsht reports them
    as HX_ERR_ARG before anything is scattered."""
    from . import _lib

    if rc == _lib.HX_ERR_ARG:
        raise ValueError(_lib.load().hx_last_error().decode(errors="replace"))
    _lib.check(rc)


_warned_unit_weights = False


def _warn_unit_weights():
    """One-time note: the reference calls hp.map2alm(use_pixel_weights=True, datapath=DATAPATH); the weight
    files are healpy data this package cannot ship, so without ``pixel_weights`` / ``ring_weights`` the
    quadrature uses unit weights (+ ``niter`` Jacobi iterations) and differs from the reference at the level of
    the HEALPix quadrature error (~1e-3 relative at l ~ 2 nside without iterations, ~1e-5 with niter=3)."""
    global _warned_unit_weights
    if not _warned_unit_weights:
        import warnings

        _warned_unit_weights = True
        warnings.warn("HipHealpixMapper.transform: no pixel_weights / ring_weights given; using unit quadrature "
                      "weights (healpy's use_pixel_weights=True files are not shipped with heracles_amd). Pass "
                      "pixel_weights= (full-sky array from healpy's weight file) for reference-equal alms.",
                      RuntimeWarning, stacklevel=3)


def _native(arr):
    """heracles/healpy.py:43-55: inputs in non-native byte order are byteswapped."""
    arr = np.asanyarray(arr)
    if arr.dtype.byteorder not in ("=", "|"):
        arr = arr.view(arr.dtype.newbyteorder("=")).byteswap()
    return arr


def ang2pix_ring(nside, lon, lat, out=None):
    """RING pixel index of (lon, lat) in degrees on the GPU: ``hp.ang2pix(nside, lon, lat,
    lonlat=True)`` of heracles/healpy.py:157.  numpy arrays or device tensors."""
    from . import _lib

    _lib.ensure_init()
    if hasattr(lon, "data_ptr"):
        import torch

        lon, lat = lon.to(torch.float64).contiguous(), lat.to(torch.float64).contiguous()
        if out is None:
            out = torch.empty(lon.shape, dtype=torch.int64, device=lon.device)
        n = lon.numel()
    else:
        lon = np.ascontiguousarray(_native(lon), dtype=np.float64)
        lat = np.ascontiguousarray(_native(lat), dtype=np.float64)
        if out is None:
            out = np.empty(lon.shape, dtype=np.int64)
        n = lon.size
    if lon.shape != lat.shape:
        raise ValueError("lon and lat must have the same shape")
    _check_points(_lib.load().hx_ang2pix_ring(int(nside), n, _lib.ptr(lon), _lib.ptr(lat), _lib.ptr(out)))
    return out


def map_values(nside, lon, lat, data, values, *, ordered=True):
    """``data[..., ipix[j]] += values[..., j]`` in catalogue order (heracles/healpy.py:58-66,
    :144-160) on the GPU; ``data`` is a numpy array (round trip through HBM) or a device
    tensor (updated in place, the path to use when looping over catalogue pages)."""
    from . import _lib

    _lib.ensure_init()
    npix = 12 * nside * nside
    if data.shape[-1] != npix:
        raise ValueError(f"maps have {data.shape[-1]} pixels, NSIDE={nside} needs {npix}")
    on_dev = hasattr(data, "data_ptr")
    if on_dev:
        import torch

        if data.dtype != torch.float64 or not data.is_contiguous():
            raise TypeError("device maps must be contiguous float64")
        maps = data
        conv = lambda a: (a if hasattr(a, "data_ptr") else torch.as_tensor(np.ascontiguousarray(_native(a), dtype=np.float64))
                          ).to(torch.float64).contiguous()
        lon, lat, values = conv(lon), conv(lat), conv(values)
        n = lon.numel()
    else:
        lon = np.ascontiguousarray(_native(lon), dtype=np.float64)
        lat = np.ascontiguousarray(_native(lat), dtype=np.float64)
        values = np.ascontiguousarray(_native(values), dtype=np.float64)
        n = lon.size
        maps = data if (data.dtype == np.float64 and data.flags.c_contiguous and data.dtype.isnative) \
            else np.ascontiguousarray(data, dtype=np.float64)
    if tuple(lat.shape) != tuple(lon.shape) or values.shape[-1] != n:
        raise ValueError("lon, lat and the last axis of values must have the same length")
    nval = 1
    for d in data.shape[:-1]:
        nval *= d
    if tuple(values.shape[:-1]) != tuple(data.shape[:-1]):
        # the compiled reference loop broadcasts values[..., j] into maps[..., i]
        values = (np.broadcast_to(values, (*data.shape[:-1], n)).copy() if not on_dev
                  else values.expand(*data.shape[:-1], n).contiguous())
    _check_points(_lib.load().hx_map_values(int(nside), n, _lib.ptr(lon), _lib.ptr(lat), nval, _lib.ptr(values),
                                            _lib.ptr(maps), 0 if ordered else 1))
    if maps is not data:
        data[...] = maps


def ud_grade(data, nside_out, dtype=np.float64):
    """``hp.ud_grade(data, nside_out, dtype=dtype)`` for RING maps (pess=False, power=None):
    mean of the unmasked children / replication, on the GPU.  numpy in -> numpy out,
    device tensor in -> device tensor out."""
    from . import _lib

    _lib.ensure_init()
    npix_in = data.shape[-1]
    nside_in = int(round((npix_in / 12) ** 0.5))
    if 12 * nside_in * nside_in != npix_in:
        raise ValueError("Wrong pixel number (it is not 12*nside**2)")
    for ns in (nside_in, nside_out):
        if ns < 1 or ns & (ns - 1):
            raise ValueError(f"{ns} is not a valid nside parameter (must be a power of 2, less than 2**30)")
    npix_out = 12 * nside_out * nside_out
    nmaps = 1
    for d in data.shape[:-1]:
        nmaps *= d
    if hasattr(data, "data_ptr"):
        import torch

        src = data.to(torch.float64).contiguous()
        out = torch.empty((*data.shape[:-1], npix_out), dtype=torch.float64, device=data.device)
    else:
        src = np.ascontiguousarray(_native(data), dtype=np.float64)
        out = np.empty((*data.shape[:-1], npix_out), dtype=np.float64)
    _lib.check(_lib.load().hx_ud_grade(nside_in, int(nside_out), nmaps, _lib.ptr(src), _lib.ptr(out)))
    if not hasattr(out, "data_ptr") and np.dtype(dtype) != np.float64:
        out = out.astype(dtype)
    return out


class HipHealpixMapper:
    """Mapper for HEALPix maps whose ``transform`` runs on MI355X."""

    #: directory of healpy's pixel-weight files, as HealpixMapper.DATAPATH (heracles/healpy.py:73; set from the configuration
    #: at heracles/cli.py:536-538): ``<DATAPATH>/full_weights/healpix_full_weights_nside_NNNN.fits`` is read on the first
    #: transform, expanded to the full sky on the GPU and kept in HBM (heracles_amd/weights.py)
    DATAPATH = None

    def __init__(self, nside, lmax=None, *, deconvolve=None, dtype=np.float64, niter=3,
                 pixwin=None, pixel_weights=None, ring_weights=None, datapath=None):
        if lmax is None:
            lmax = 3 * nside // 2
        if deconvolve is None:
            deconvolve = True
        self.__nside = nside
        self.__lmax = lmax
        self.__deconv = deconvolve
        self.__dtype = np.dtype(dtype)
        #: Jacobi iterations; healpy.map2alm's default is iter=3 and the reference passes none
        self.niter = niter
        self.pixwin = pixwin
        self.pixel_weights = pixel_weights
        self.ring_weights = ring_weights
        self.datapath = datapath

    @property
    def nside(self):
        return self.__nside

    @property
    def lmax(self):
        return self.__lmax

    @property
    def deconvolve(self):
        return self.__deconv

    @property
    def area(self):
        """Pixel area in steradians (hp.nside2pixarea, heracles/healpy.py:117-122)."""
        return 4.0 * np.pi / (12 * self.__nside**2)

    def create(self, *dims, spin=0):
        m = np.zeros((*dims, 12 * self.__nside**2), dtype=self.__dtype)
        update_metadata(m, geometry="healpix", kernel="healpix", nside=self.__nside,
                        lmax=self.__lmax, deconv=self.__deconv, spin=spin)
        return m

    def map_values(self, lon, lat, data, values, spin=0):
        """Add values to the pixels containing (lon, lat) [degrees]; heracles/healpy.py:144-160.
        ang2pix, the stable sort by pixel and the ordered per-pixel sums run on the GPU."""
        map_values(self.__nside, lon, lat, data, values)

    def _fl(self, spin):
        if not self.__deconv:
            return None
        pw0, pw2 = pixel_window(self.__nside, self.__lmax, self.pixwin)
        pw = pw0 if spin == 0 else pw2
        fl = np.ones(self.__lmax + 1)
        fl[abs(spin):] /= pw[abs(spin):]
        return fl

    def _load_weights(self):
        """use_pixel_weights=True, datapath=DATAPATH of heracles/healpy.py:183-189: the weight file of this resolution, if a
        data path is configured and holds one."""
        if self.pixel_weights is None and self.ring_weights is None:
            path = self.datapath if self.datapath is not None else type(self).DATAPATH
            if path is not None:
                from .weights import load_pixel_weights, weights_filename

                self.pixel_weights = load_pixel_weights(path, self.__nside)
                if self.pixel_weights is None:
                    # healpy raises when use_pixel_weights=True finds no file under datapath; silently falling back to unit weights
                    # would change the quadrature behind the caller's back
                    raise FileNotFoundError(f"no pixel-weight file {weights_filename(self.__nside)} under datapath {path!r}")

    def transform(self, data, spin=0):
        """Spherical harmonic transform of HEALPix maps; heracles/healpy.py:162-203."""
        if spin not in (0, 2):
            raise NotImplementedError(f"spin-{spin} maps not yet supported")
        fl = self._fl(spin)
        self._load_weights()
        if self.pixel_weights is None and self.ring_weights is None:
            _warn_unit_weights()
        plan = sht.get_plan(self.__nside, self.__lmax)
        if hasattr(data, "data_ptr"):
            # device-resident maps (e.g. accumulated by map_values on the GPU): alms stay in HBM; a torch
            # tensor cannot carry dtype metadata, so none is attached
            if spin == 2 and (data.ndim < 2 or data.shape[-2] != 2):
                raise ValueError("spin-2 maps must have shape (..., 2, npix)")
            return plan.map2alm(data, spin, ring_weights=self.ring_weights, pix_weights=self.pixel_weights,
                                fl=fl, niter=self.niter)
        md = data.dtype.metadata or {}
        maps = np.ascontiguousarray(_native(data), dtype=np.float64)
        if spin == 2 and (maps.ndim < 2 or maps.shape[-2] != 2):
            raise ValueError("spin-2 maps must have shape (..., 2, npix)")
        alm = plan.map2alm(maps, spin, ring_weights=self.ring_weights,
                           pix_weights=self.pixel_weights, fl=fl, niter=self.niter)
        update_metadata(alm, **{**md, "deconv": self.__deconv})
        return alm

    def transform_many(self, maps, spins, *, device=None):
        """Batched transform of a list of maps (the loop of heracles/mapping.py:151-172 as one call): the arrays go to
        ``hx_map2alm_list`` as they are -- no stacked copy on the host, one upload pipeline across spins (``niter = 0``) or one
        resident batch per spin (``niter > 0``).  ``device="cuda"``: the alms stay in HBM and come back as ``DeviceArray``s (what
        ``angular_power_spectra`` takes without another PCIe round trip); default: numpy arrays, as the reference returns."""
        for sp in spins:
            if sp not in (0, 2):
                raise NotImplementedError(f"spin-{sp} maps not yet supported")
        self._load_weights()
        if self.pixel_weights is None and self.ring_weights is None:
            _warn_unit_weights()
        plan = sht.get_plan(self.__nside, self.__lmax)
        npix, nlm = 12 * self.__nside**2, (self.__lmax + 1) * (self.__lmax + 2) // 2
        native = [m if hasattr(m, "data_ptr") else np.ascontiguousarray(_native(m), dtype=np.float64) for m in maps]
        native = [m.reshape((npix,) if sp == 0 else (2, npix)) for m, sp in zip(native, spins)]
        outs = None
        if device is not None:
            import torch

            outs = [torch.empty((nlm,) if sp == 0 else (2, nlm), dtype=torch.complex128, device=device) for sp in spins]
        alms = plan.map2alm_list(native, spins, outs=outs, ring_weights=self.ring_weights, pix_weights=self.pixel_weights,
                                 fl0=self._fl(0), fl2=self._fl(2), niter=self.niter)
        out = []
        for m, a, sp in zip(maps, alms, spins):
            md = {**((m.dtype.metadata or {}) if isinstance(m, np.ndarray) else {"spin": sp}), "deconv": self.__deconv}
            if device is not None:
                out.append(DeviceArray(a, md))
                continue
            if hasattr(a, "data_ptr"):
                # a device-resident map with device=None: the alm stays in HBM as a tensor, as transform() returns it (a torch tensor
                # cannot carry dtype metadata)
                out.append(a)
                continue
            a = a if isinstance(a, np.ndarray) else np.array(a)
            update_metadata(a, **md)
            out.append(a)
        return out

    def resample(self, data):
        """Change resolution of HEALPix maps (hp.ud_grade, heracles/healpy.py:205-209) on the GPU."""
        return ud_grade(data, self.__nside, dtype=self.__dtype)
