"""Mapper that holds alms directly: the ``heracles.ducc.DiscreteMapper`` surface
(heracles/ducc.py:40-162) in front of the same two-point backend.

``create`` / ``transform`` / ``resample`` / ``area`` are reproduced;
``resample`` re-packs on the GPU through ``hx_alm_resample``.  ``map_values`` -- ducc0's
non-uniform adjoint synthesis (heracles/ducc.py:92-133) -- is a separate kernel family
(NUFFT + SHT) and is not provided: it raises, it does not fall back.
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
        raise NotImplementedError(
            "HipDiscreteMapper.map_values (ducc0 adjoint_synthesis_general, heracles/ducc.py:92-133) "
            "has no HIP implementation yet; there is no CPU fallback"
        )

    def transform(self, data, spin=0):
        """Does nothing, since inputs are alms already (heracles/ducc.py:135-143)."""
        return data

    def resample(self, data):
        """Change LMAX of alm (heracles/ducc.py:145-162)."""
        return alm_resample(data, self.__lmax, dtype=self.__dtype)
