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


def _cutoff_exponent(wm, theta_max):
    """log10 of the mask correlation below which it is damped: 1e-5, or its value at the Gauss-Legendre node closest
    to ``theta_max`` degrees in the first mask spectrum (heracles/unmixing.py:83-91)."""
    if theta_max is None:
        return -5
    first = next(iter(wm.values()))
    nodes, _ = gauss_legendre(first.shape[first.axis[0]])
    nearest = np.abs(np.degrees(np.arccos(nodes)) - theta_max).argmin()
    return np.log10(abs(first[nearest]))


def _damp_in_place(xi_mask, x0):
    """xi_m <- xi_m (1 + exp(-50 (log10 |xi_m| - x0))), written through to the caller's array: the reference modifies
    the object its lookup returns (heracles/unmixing.py:99), so a mask pair shared by two data keys is damped twice."""
    xi_mask *= logistic(np.log10(abs(xi_mask)), x0=x0)
    return xi_mask


def _naturalspice(wd, wm, fields, theta_max=None):
    """xi_d / damped xi_m for every data key (a, b, i, j) with the masks of fields a and b (heracles/unmixing.py:66-102)."""
    mask_of = {name: f.mask for name, f in fields.items() if f.mask is not None}
    x0 = _cutoff_exponent(wm, theta_max)

    def corrected(key):
        a, b, i, j = key
        xi_mask = _damp_in_place(_get_cl((mask_of[a], mask_of[b], i, j), wm).array, x0)
        return replace(wd[key], array=wd[key].array / xi_mask)

    return {key: corrected(key) for key in wd}


def naturalspice(d, m, fields, theta_max=None):
    """Natural unmixing of data spectra d by mask spectra m (heracles/unmixing.py:36-64): both go to correlation
    functions on the mask's band limit, the ratio comes back and is cut to the data's band limit."""
    def band_limit(spectra):
        first = next(iter(spectra.values()))
        return first.shape[first.axis[0]]

    n_data, n_mask = band_limit(d), band_limit(m)
    ratio = _naturalspice(cl2corr(_pad(d, n_mask)), cl2corr(m), fields, theta_max=theta_max)
    return _pad(corr2cl(ratio), n_data)
