"""Mapper that holds alms directly: the ``heracles.ducc.DiscreteMapper`` surface
(heracles/ducc.py:40-162) in front of the same two-point backend.

``create`` / ``transform`` / ``resample`` / ``area`` are reproduced;
``resample`` re-packs on the GPU through ``hx_alm_resample``.  ``map_values`` -- ducc0's
non-uniform adjoint synthesis (heracles/ducc.py:92-133) -- runs through
``hx_pointsht_adjoint``: a type-1 non-uniform FFT on the (theta, phi) torus and the Legendre
analysis kernel of the HEALPix path on equidistant rings (csrc/hx_nufft.hip).
"""

from __future__ import annotations

import numpy as np

from . import _lib
from .core import update_metadata


def alm_resample(data, lmax_out, dtype=np.complex128):
    """Change the band limit of m-major alms (..., nlm_in) -> (..., nlm_out): truncate or zero-pad."""
    _lib.ensure_init()
    n = data.shape[-1]
    lmax_in = (int((8 * n + 1) ** 0.5 + 0.01) - 3) // 2
    if (lmax_in + 1) * (lmax_in + 2) // 2 != n:
        raise ValueError(f"{n} is not a triangular alm size")
    n_out = (lmax_out + 1) * (lmax_out + 2) // 2
    ncomp = 1
    for d in data.shape[:-1]:
        ncomp *= d
    if hasattr(data, "data_ptr"):
        import torch

        src = data.to(torch.complex128).contiguous()
        out = torch.empty((*data.shape[:-1], n_out), dtype=torch.complex128, device=data.device)
    else:
        src = np.ascontiguousarray(data, dtype=np.complex128)
        out = np.empty((*data.shape[:-1], n_out), dtype=np.complex128)
    _lib.check(_lib.load().hx_alm_resample(lmax_in, int(lmax_out), ncomp, _lib.ptr(src), _lib.ptr(out)))
    if not hasattr(out, "data_ptr") and np.dtype(dtype) != np.complex128:
        out = out.astype(dtype)
    return out


class PointSHT:
    """``alm = sum_p values_p conj(sY_lm(theta_p, phi_p))`` for points anywhere on the sphere: the object behind
    ``hx_pointsht_*`` (one per band limit and accuracy; it owns the oversampled grid and the ring plan)."""

    def __init__(self, lmax, epsilon=1e-12):
        _lib.ensure_init()
        self.lmax, self.epsilon = int(lmax), float(epsilon)
        self._h = _lib.load().hx_pointsht_create(self.lmax, self.epsilon)
        if not self._h:
            raise _lib.HxError(-1, _lib.load().hx_last_error().decode(errors="replace"))
        import ctypes

        info = (ctypes.c_int * 4)()
        _lib.check(_lib.load().hx_pointsht_info(self._h, info))
        _, self.nrings_circle, self.ngrid, self.kernel_width = list(info)
        self.nlm = (self.lmax + 1) * (self.lmax + 2) // 2

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().hx_pointsht_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def adjoint_synthesis(self, loc, values, spin=0, out=None):
        """values (ncomp, npoints) float64 at loc (npoints, 2) = (colatitude, longitude) [rad] -> alm (ncomp, nlm)
        complex128; numpy in -> numpy out, torch device tensors in -> device tensor out."""
        dev = hasattr(values, "data_ptr")
        if dev:
            import torch

            values = values.to(torch.float64).contiguous()
            loc = loc.to(torch.float64).contiguous()
            if out is None:
                out = torch.empty((values.shape[0], self.nlm), dtype=torch.complex128, device=values.device)
        else:
            values = np.ascontiguousarray(values, dtype=np.float64)
            loc = np.ascontiguousarray(loc, dtype=np.float64)
            if out is None:
                out = np.empty((values.shape[0], self.nlm), dtype=np.complex128)
        if values.ndim != 2 or tuple(loc.shape) != (values.shape[1], 2):
            raise ValueError("values must be (ncomp, npoints) and loc (npoints, 2)")
        try:
            _lib.check(_lib.load().hx_pointsht_adjoint(self._h, int(spin), int(values.shape[0]), int(values.shape[1]),
                                                       _lib.ptr(loc), _lib.ptr(values), _lib.ptr(out)))
        except _lib.HxError as e:
            if e.code == _lib.HX_ERR_ARG:
                raise ValueError(str(e)) from None
            raise
        return out


_point_plans = {}


def get_point_sht(lmax, epsilon=1e-12):
    key = (int(lmax), float(epsilon))
    if key not in _point_plans:
        _point_plans[key] = PointSHT(*key)
    return _point_plans[key]


class HipDiscreteMapper:
    """Mapper that creates alms directly (heracles/ducc.py:40-162)."""

    def __init__(self, lmax, *, dtype=np.complex128, nthreads=0):
        self.__lmax = lmax
        self.__dtype = np.dtype(dtype)
        self.__nthreads = nthreads  # accepted for signature compatibility; unused

    @property
    def lmax(self):
        return self.__lmax

    @property
    def area(self):
        """The effective area for this mapper (heracles/ducc.py:66-71)."""
        return 1.0

    def create(self, *dims, spin=0):
        lmax = self.__lmax
        m = np.zeros((*dims, (lmax + 1) * (lmax + 2) // 2), dtype=self.__dtype)
        update_metadata(m, geometry="discrete", kernel="none", lmax=lmax, spin=spin)
        return m

    def map_values(self, lon, lat, data, values, spin=0):
        """Add values to alms (heracles/ducc.py:92-133): ``data += sum_p values_p conj(sY_lm(lon_p, lat_p))``.
        float32 values are transformed to 1e-5, everything else in float64 to 1e-12, as the reference asks of ducc0."""
        values = np.asarray(values)
        flatten = values.ndim == 1
        if flatten:
            values = values.reshape(1, -1)
        epsilon = 1e-5 if values.dtype == np.float32 else 1e-12
        lon, lat = np.asarray(lon, dtype=np.float64), np.asarray(lat, dtype=np.float64)
        loc = np.empty((lon.size, 2), dtype=np.float64)
        loc[:, 0] = np.radians(90.0 - lat)
        loc[:, 1] = np.radians(lon % 360.0)
        alms = get_point_sht(self.__lmax, epsilon).adjoint_synthesis(loc, values, spin=spin)
        if flatten:
            alms = alms[0]
        data += alms

    def transform(self, data, spin=0):
        """Does nothing, since inputs are alms already (heracles/ducc.py:135-143)."""
        return data

    def resample(self, data):
        """Change LMAX of alm (heracles/ducc.py:145-162)."""
        return alm_resample(data, self.__lmax, dtype=self.__dtype)
