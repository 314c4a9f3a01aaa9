"""Plan-level interface to the HIP spherical harmonic transforms (libhxsht)."""

from __future__ import annotations

import numpy as np

from . import _lib


def nlm(lmax: int) -> int:
    return (lmax + 1) * (lmax + 2) // 2


def _is_tensor(a) -> bool:
    return hasattr(a, "data_ptr") and not isinstance(a, np.ndarray)


class Plan:
    """Ring tables, recursion tables and device scratch for one (nside, lmax).

    Replaces the implicit plan inside ``healpy.map2alm`` (heracles/healpy.py:183-189).
    Inputs may be numpy arrays (host; staged by the library) or contiguous torch CUDA
    tensors of dtype float64 / complex128 (used in place, results stay in HBM).
    """

    def __init__(self, nside: int, lmax: int, max_comp: int = 8):
        L = _lib.load()
        _lib.ensure_init()
        self.nside, self.lmax = int(nside), int(lmax)
        self.npix = 12 * self.nside**2
        self.nlm = nlm(self.lmax)
        self._h = L.hx_plan_create(self.nside, self.lmax, int(max_comp))
        if not self._h:
            raise _lib.HxError(-1, L.hx_last_error().decode(errors="replace"))

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().hx_plan_destroy(self._h)
            self._h = None

    def release_scratch(self):
        """Free the plan's transient HBM scratch (hx_plan_release_scratch: up to ~200 GB after a full-size job); it is allocated again on
        demand."""
        _lib.check(_lib.load().hx_plan_release_scratch(self._h))

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    @property
    def scratch_bytes(self) -> int:
        return int(_lib.load().hx_plan_scratch_bytes(self._h))

    @property
    def last_chunks(self) -> int:
        """m-chunks of the most recent analysis sweep (see hx_set_scratch_budget)."""
        return int(_lib.load().hx_plan_last_chunks(self._h))

    def mfma_flops(self, spin, ncomp):
        """Matrix-instruction flops one map2alm(niter=0) of ncomp components executes."""
        import ctypes

        out = ctypes.c_double(0.0)
        _lib.check(_lib.load().hx_plan_mfma_flops(self._h, int(spin), int(ncomp), ctypes.byref(out)))
        return out.value

    def executed_flops(self, spin, ncomp):
        """(matrix-instruction flops, FP64 vector flops of the recursions) one map2alm(niter=0) executes."""
        import ctypes

        out = (ctypes.c_double * 2)()
        _lib.check(_lib.load().hx_plan_executed_flops(self._h, int(spin), int(ncomp), out))
        return out[0], out[1]

    # -- helpers ------------------------------------------------------------------
    def _out_like(self, ref, shape, complex_):
        if _is_tensor(ref):
            import torch

            return torch.empty(shape, dtype=torch.complex128 if complex_ else torch.float64,
                               device=ref.device)
        return np.empty(shape, dtype=np.complex128 if complex_ else np.float64)

    @staticmethod
    def _prep(a, dtype):
        if a is None:
            return None
        if _is_tensor(a):
            return a.contiguous()
        return np.ascontiguousarray(a, dtype=dtype)

    # -- transforms ---------------------------------------------------------------
    def map2alm(self, maps, spin=0, *, ring_weights=None, pix_weights=None, fl=None,
                niter=0, out=None):
        """maps (..., npix) -> alm (..., nlm).  spin 2: the leading axis pairs (Q, U) -> (E, B)."""
        maps = self._prep(maps, np.float64)
        lead = tuple(maps.shape[:-1])
        if maps.shape[-1] != self.npix:
            raise ValueError(f"map has {maps.shape[-1]} pixels, plan expects {self.npix}")
        ncomp = int(np.prod(lead)) if lead else 1
        if out is None:
            out = self._out_like(maps, lead + (self.nlm,), True)
        rw = self._prep(ring_weights, np.float64)
        pw = self._prep(pix_weights, np.float64)
        flv = self._prep(fl, np.float64)
        if flv is not None and flv.shape[-1] != self.lmax + 1:
            raise ValueError("fl must have lmax+1 entries")
        _lib.check(_lib.load().hx_map2alm(self._h, int(spin), ncomp, _lib.ptr(maps), _lib.ptr(out),
                                          _lib.ptr(rw), _lib.ptr(pw), _lib.ptr(flv), int(niter)))
        return out

    def map2alm_multi(self, jobs, *, ring_weights=None, pix_weights=None):
        """Several transforms in one call: ``jobs`` = [(maps, spin, out_or_None[, fl])]; host maps of all jobs share one upload
        pipeline (uploads of the next sweep overlap the transform of the current one across jobs).  Returns the list of alms.
        Put the large jobs first: what stays exposed behind the last uploaded byte is the transform of the last sweep."""
        import ctypes as C

        n = len(jobs)
        keep, outs = [], []
        spins, ncomps = (C.c_int * n)(), (C.c_int * n)()
        pm, pa, pf = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
        for j, job in enumerate(jobs):
            maps, spin, out = job[0], job[1], job[2]
            fl = job[3] if len(job) > 3 else None
            maps = self._prep(maps, np.float64)
            if maps.shape[-1] != self.npix:
                raise ValueError(f"map has {maps.shape[-1]} pixels, plan expects {self.npix}")
            lead = tuple(maps.shape[:-1])
            nc = int(np.prod(lead)) if lead else 1
            if out is None:
                out = self._out_like(maps, lead + (self.nlm,), True)
            flv = self._prep(fl, np.float64)
            if flv is not None and flv.shape[-1] != self.lmax + 1:
                raise ValueError("fl must have lmax+1 entries")
            keep += [maps, flv]
            outs.append(out)
            spins[j], ncomps[j] = int(spin), nc
            pm[j], pa[j], pf[j] = _lib.ptr(maps), _lib.ptr(out), _lib.ptr(flv)
        rw = self._prep(ring_weights, np.float64)
        pw = self._prep(pix_weights, np.float64)
        _lib.check(_lib.load().hx_map2alm_multi(self._h, n, spins, ncomps, pm, pa, _lib.ptr(rw), _lib.ptr(pw), pf))
        return outs

    def map2alm_list(self, maps, spins, *, outs=None, ring_weights=None, pix_weights=None, fl0=None, fl2=None, niter=0):
        """The transform loop of ``heracles.transform`` (heracles/mapping.py:151-172) as one call over SEPARATE arrays: ``maps[i]``
        is ``(npix,)`` for spin 0 or ``(2, npix)`` (Q, U) for spin 2, numpy or device; returns one alm array per map (``(nlm,)`` /
        ``(2, nlm)``).  The maps are gathered into the upload pipeline sweep by sweep (no stacked host copy), spin-2 fields first;
        with ``niter > 0`` the maps of a spin are gathered into one device array and iterated as a batch."""
        import ctypes as C

        n = len(maps)
        keep, res = [], []
        sp = (C.c_int * n)()
        pm, pa = (C.c_void_p * n)(), (C.c_void_p * n)()
        for i, (m, s) in enumerate(zip(maps, spins)):
            m = self._prep(m, np.float64)
            want = (self.npix,) if int(s) == 0 else (2, self.npix)
            if int(s) not in (0, 2):
                raise NotImplementedError(f"spin-{s} maps not yet supported")
            if tuple(m.shape) != want:
                raise ValueError(f"map {i} of spin {s} has shape {tuple(m.shape)}, expected {want}")
            out = outs[i] if outs is not None else self._out_like(m, want[:-1] + (self.nlm,), True)
            keep.append(m)
            res.append(out)
            sp[i] = int(s)
            pm[i], pa[i] = _lib.ptr(m), _lib.ptr(out)
        rw = self._prep(ring_weights, np.float64)
        pw = self._prep(pix_weights, np.float64)
        f0, f2 = self._prep(fl0, np.float64), self._prep(fl2, np.float64)
        for f in (f0, f2):
            if f is not None and f.shape[-1] != self.lmax + 1:
                raise ValueError("fl must have lmax+1 entries")
        _lib.check(_lib.load().hx_map2alm_list(self._h, n, sp, pm, pa, _lib.ptr(rw), _lib.ptr(pw), _lib.ptr(f0), _lib.ptr(f2), int(niter)))
        return res

    def alm2map(self, alms, spin=0, *, out=None):
        alms = self._prep(alms, np.complex128)
        lead = tuple(alms.shape[:-1])
        if alms.shape[-1] != self.nlm:
            raise ValueError(f"alm has {alms.shape[-1]} coefficients, plan expects {self.nlm}")
        ncomp = int(np.prod(lead)) if lead else 1
        if out is None:
            out = self._out_like(alms, lead + (self.npix,), False)
        _lib.check(_lib.load().hx_alm2map(self._h, int(spin), ncomp, _lib.ptr(alms), _lib.ptr(out)))
        return out


_plans: dict = {}


def get_plan(nside: int, lmax: int) -> Plan:
    """Process-wide plan cache (tables for nside=4096 take a few hundred MB of HBM)."""
    key = (int(nside), int(lmax))
    if key not in _plans:
        _plans[key] = Plan(nside, lmax)
    return _plans[key]


def clear_plans():
    for p in _plans.values():
        p.close()
    _plans.clear()
