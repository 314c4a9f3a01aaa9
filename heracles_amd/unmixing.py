"""NaturalSpice real-space unmixing on top of the GPU Cl <-> xi transforms.

Mirrors heracles/unmixing.py:32-102.  The reference regularises the mask correlation
function IN PLACE through the object returned by ``get_cl`` (unmixing.py:99), so a mask
pair shared by two data keys is regularised twice; that behaviour is kept.
"""

from __future__ import annotations

from dataclasses import replace

import numpy as np

from .transforms import cl2corr, corr2cl, gauss_legendre


def logistic(x, x0=-2, k=50):
    return 1.0 + np.exp(-k * (x - x0))


def _get_cl(key, cls):
    """Symmetric lookup with spin/axis swap (heracles/utils.py:28-52)."""
    if key in cls:
        return cls[key]
    a, b, i, j = key
    sym = (b, a, j, i)
    if sym not in cls:
        raise KeyError(f"Key {key} not found in Cls.")
    arr = cls[sym].array
    s1, s2 = cls[sym].spin
    if s1 != 0 and s2 != 0:
        arr = np.transpose(arr, axes=(1, 0, 2))
    return replace(cls[sym], array=arr, spin=(s2, s1))


def _pad(d, n):
    """Zero-pad / truncate spectra to n multipoles; equals binned(d, arange(0, n+1)) of
    heracles/unmixing.py:53,63 for unit-width bins."""
    out = {}
    for key, r in d.items():
        a = np.asarray(r.array)
        m = a.shape[-1]
        if m >= n:
            b = np.array(a[..., :n])
        else:
            b = np.concatenate([a, np.zeros(a.shape[:-1] + (n - m,), dtype=a.dtype)], axis=-1)
        out[key] = replace(r, array=b, ell=np.arange(n), lower=np.arange(n), upper=np.arange(1, n + 1),
                           weight=np.ones(n))
    return out


def _naturalspice(wd, wm, fields, theta_max=None):
    masks = {k: f.mask for k, f in fields.items() if f.mask is not None}
    if theta_max is not None:
        first = list(wm.values())[0]
        lmax_mask = first.shape[first.axis[0]]
        xvals, _ = gauss_legendre(lmax_mask)
        theta = np.arccos(xvals) * 180 / np.pi
        i_max = np.abs(theta - theta_max).argmin()
        x0 = np.log10(abs(first[i_max]))
    else:
        x0 = -5
    out = {}
    for key in wd:
        a, b, i, j = key
        _wm = _get_cl((masks[a], masks[b], i, j), wm).array
        _wd = wd[key].array
        _wm *= logistic(np.log10(abs(_wm)), x0=x0)
        out[key] = replace(wd[key], array=(_wd / _wm))
    return out


def naturalspice(d, m, fields, theta_max=None):
    """Natural unmixing of data spectra d by mask spectra m (heracles/unmixing.py:36-64)."""
    first_wd = list(d.values())[0]
    first_wm = list(m.values())[0]
    lmax = first_wd.shape[first_wd.axis[0]]
    lmax_mask = first_wm.shape[first_wm.axis[0]]
    d = _pad(d, lmax_mask)
    wd = cl2corr(d)
    wm = cl2corr(m)
    corr = _naturalspice(wd, wm, fields, theta_max=theta_max)
    return _pad(corr2cl(corr), lmax)
