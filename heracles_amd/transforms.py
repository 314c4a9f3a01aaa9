"""Cl <-> correlation function transforms on the GPU.

Mirrors heracles/transforms.py: ``_cl2corr`` / ``_corr2cl`` (transforms.py:115-204) and the
dict-level ``cl2corr`` / ``corr2cl`` (transforms.py:207-363).  The reference loops over
lmax+1 Gauss-Legendre nodes in Python; here all nodes and all spectra go in one launch.
"""

from __future__ import annotations

from dataclasses import replace

import numpy as np

from . import _lib

_gl_cache: dict = {}


def gauss_legendre(n: int):
    """Nodes and weights; usable as the reference's ``transforms.gauss_legendre`` hook."""
    if n not in _gl_cache:
        x = np.empty(n)
        w = np.empty(n)
        _lib.ensure_init()
        _lib.check(_lib.load().hx_gauss_legendre(int(n), _lib.ptr(x), _lib.ptr(w)))
        x.flags.writeable = False
        w.flags.writeable = False
        _gl_cache[n] = (x, w)
    return _gl_cache[n]


def wigner_d_table(lmax, a, b, x):
    """d^l_{ab}(x_k) for l=0..lmax at all x: array (len(x), lmax+1)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty((x.shape[0], lmax + 1))
    _lib.ensure_init()
    _lib.check(_lib.load().hx_wigner_d_table(int(lmax), int(a), int(b), x.shape[0], _lib.ptr(x), _lib.ptr(out)))
    return out


def legendre_funcs(lmax, x, m=(0, 2), lfacs=None, lfacs2=None, lrootfacs=None):
    """``heracles.transforms.legendre_funcs`` (heracles/transforms.py:46-112) for a scalar ``x``: the list
    ``[(P, P'), (d11, dm11), (d20, d22, d2m2)]`` restricted to the requested ``m``; P starts at l = 0, the spin-1 arrays at
    l = 1, the spin-2 arrays at l = 2.  The values come from the Wigner-d tables of the mixing-matrix kernels (three-term
    recursions in l, stable also where the reference switches to its small-angle series); ``lfacs*`` are accepted and ignored."""
    xs = np.array([float(x)])
    res = []
    if 0 in m:
        P = wigner_d_table(lmax, 0, 0, xs)[0]
        dP = np.zeros(lmax + 1)
        # P'_l = P'_{l-2} + (2l - 1) P_{l-1}
        inc = (2.0 * np.arange(1, lmax + 1) - 1.0) * P[:-1]
        dP[1::2] = np.cumsum(inc[0::2])
        dP[2::2] = np.cumsum(inc[1::2])
        res.append((P, dP))
    if 1 in m:
        res.append((wigner_d_table(lmax, 1, 1, xs)[0, 1:], wigner_d_table(lmax, -1, 1, xs)[0, 1:]))
    if 2 in m:
        res.append((wigner_d_table(lmax, 2, 0, xs)[0, 2:], wigner_d_table(lmax, 2, 2, xs)[0, 2:], wigner_d_table(lmax, 2, -2, xs)[0, 2:]))
    return res


def _as4(arr):
    arr = np.asarray(arr, dtype=np.float64)
    if arr.ndim == 1:
        z = np.zeros_like(arr)
        arr = np.array([arr, z, z, z]).T
    return arr


def _cl2corr(cls, lmax=None, sampling_factor=1):
    """(L,4) [TT,EE,BB,TE] -> (N,4) [T, Q+U, Q-U, cross] at N=lmax+1 GL nodes."""
    if sampling_factor != 1:
        raise NotImplementedError("sampling_factor != 1")
    cls = _as4(cls)
    if lmax is None:
        lmax = cls.shape[0] - 1
    c = np.ascontiguousarray(cls[: lmax + 1])[None]
    out = np.empty_like(c)
    _lib.ensure_init()
    _lib.check(_lib.load().hx_cl2corr(int(lmax), 1, _lib.ptr(c), _lib.ptr(out)))
    return out[0]


def _corr2cl(corrs, lmax=None, sampling_factor=1):
    if sampling_factor != 1:
        raise NotImplementedError("sampling_factor != 1")
    corrs = _as4(corrs)
    if lmax is None:
        lmax = corrs.shape[0] - 1
    c = np.ascontiguousarray(corrs)[None]
    out = np.empty((1, lmax + 1, 4))
    _lib.ensure_init()
    _lib.check(_lib.load().hx_corr2cl(int(lmax), 1, _lib.ptr(c), _lib.ptr(out)))
    return out[0]


def _batch(fn, specs, lmax):
    """Run cl2corr/corr2cl on a list of (L,4) arrays in one launch."""
    c = np.ascontiguousarray(np.stack(specs))
    out = np.empty_like(c)
    _lib.ensure_init()
    _lib.check(fn(int(lmax), c.shape[0], _lib.ptr(c), _lib.ptr(out)))
    return out


def _ell_len(res):
    ell = getattr(res, "ell", None)
    if ell is None:
        return res.shape[res.axis[0]]
    if isinstance(ell, tuple):
        ell = ell[0]
    return len(ell)


def cl2corr(cls):
    """Dict of Result spectra -> dict of correlation functions (transforms.py:207-283)."""
    L = _lib.load()
    work, plan = [], []
    for key, cl in cls.items():
        s1, s2 = cl.spin
        lmax = _ell_len(cl) - 1
        a = np.asarray(cl.array)
        z = np.zeros(lmax + 1)
        if s1 != 0 and s2 != 0:
            specs = [np.array([z, a[0, 0], a[1, 1], z]).T, np.array([z, -a[0, 1], a[1, 0], z]).T]
        elif s1 != 0 or s2 != 0:
            specs = [np.array([z, z, z, a[0] + a[1]]).T, np.array([z, z, z, a[0] - a[1]]).T]
        else:
            specs = [np.array([a, z, z, z]).T]
        plan.append((key, lmax, len(work), len(specs)))
        work.extend((lmax, s) for s in specs)
    outs = [None] * len(work)
    for lm in sorted({w[0] for w in work}):
        ids = [i for i, w in enumerate(work) if w[0] == lm]
        res = _batch(L.hx_cl2corr, [work[i][1] for i in ids], lm)
        for i, r in zip(ids, res):
            outs[i] = r
    wds = {}
    for key, lmax, start, n in plan:
        cl = cls[key]
        s1, s2 = cl.spin
        xvals, _ = gauss_legendre(lmax + 1)
        a = np.asarray(cl.array)
        wd = np.zeros_like(a)
        if s1 != 0 and s2 != 0:
            r, i = outs[start].T, outs[start + 1].T
            wd[0, 0], wd[1, 1], wd[0, 1], wd[1, 0] = r[1], r[2], i[1], i[2]
        elif s1 != 0 or s2 != 0:
            wd[0], wd[1] = outs[start].T[3], outs[start + 1].T[3]
        else:
            wd = outs[start].T[0]
        wd = np.array(list(wd), dtype=a.dtype)
        wds[key] = replace(cl, ell=xvals, array=wd)
    return wds


def corr2cl(wds):
    """Dict of correlation functions -> dict of spectra (transforms.py:286-363)."""
    L = _lib.load()
    work, plan = [], []
    for key, wd in wds.items():
        s1, s2 = wd.spin
        lmax = _ell_len(wd) - 1
        a = np.asarray(wd.array)
        z = np.zeros(lmax + 1)
        if s1 != 0 and s2 != 0:
            specs = [np.array([z, a[0, 0], a[1, 1], z]).T, np.array([z, a[0, 1], a[1, 0], z]).T]
        elif s1 != 0 or s2 != 0:
            specs = [np.array([z, z, z, a[0]]).T, np.array([z, z, z, a[1]]).T]
        else:
            specs = [np.array([a, z, z, z]).T]
        plan.append((key, lmax, len(work), len(specs)))
        work.extend((lmax, s) for s in specs)
    outs = [None] * len(work)
    for lm in sorted({w[0] for w in work}):
        ids = [i for i, w in enumerate(work) if w[0] == lm]
        res = _batch(L.hx_corr2cl, [work[i][1] for i in ids], lm)
        for i, r in zip(ids, res):
            outs[i] = r
    cls = {}
    for key, lmax, start, n in plan:
        wd = wds[key]
        s1, s2 = wd.spin
        a = np.asarray(wd.array)
        cl = np.zeros_like(a)
        if s1 != 0 and s2 != 0:
            r, i = outs[start].T, outs[start + 1].T
            cl[0, 0], cl[1, 1], cl[0, 1], cl[1, 0] = r[1], r[2], -i[1], i[2]
        elif s1 != 0 or s2 != 0:
            p, m = outs[start].T[3], outs[start + 1].T[3]
            cl[0], cl[1] = (p + m) / 2, (p - m) / 2
        else:
            cl = outs[start].T[0]
        cl = np.array(list(cl), dtype=a.dtype)
        cls[key] = replace(wd, ell=np.arange(lmax + 1), array=cl)
    return cls
