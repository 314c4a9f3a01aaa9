"""Binning of results along their angular axes: ``heracles.result.binned`` (heracles/result.py:124-248) as the two-point
drivers use it (``angular_power_spectra(bins=, weights=)`` heracles/twopoint.py:283-284, ``mixing_matrices(bins=, weights=)``
heracles/twopoint.py:391-397, both fed by ``cli.py:696-716`` from the ``bins = 32 log 2l+1`` line of a configuration).

The rule, per angular axis: every multipole l of the axis falls into the interval of ``bins`` that ``np.digitize`` names
(``bins[i-1] <= l < bins[i]``; multipoles below the first or from the last edge on belong to no bin), carries the weight
``w_l`` = (the weight asked for: ``None`` -> 1, ``"l(l+1)"``, ``"2l+1"``, or an array) x (the weight the result already has), and a
bin holds ``sum_l w_l x_l / sum_l w_l`` -- exactly 0 where the numerator is exactly 0 (empty bins included).

``BinPlan`` is that rule for one axis as plain arrays (which bin, which weight, the normalisation), so that the binned mixing
matrices can be formed ON THE GPU from binned Wigner-d tables (``twopoint.MixmatContext.set_bins``) instead of binning a full
(3, L+1, L+1) matrix on the host; ``binned`` applies it on the host to anything else (spectra are O(lmax) per block).
"""

from __future__ import annotations

from collections.abc import Mapping

import numpy as np

from .core import Result

_WEIGHT_RULES = {
    "l(l+1)": lambda ell: ell * (ell + 1),
    "2l+1": lambda ell: 2 * ell + 1,
}


def _ratio(num, den):
    """num / den where num is not exactly zero, 0 elsewhere (the reference's ``norm``, heracles/result.py:132-135)."""
    num = np.asarray(num, dtype=float)
    out = np.zeros(np.broadcast(num, den).shape)
    np.divide(num, den, out=out, where=(num != 0))
    return out


def _axis_arrays(result, name, axes):
    """The named angular array of every ell axis of a result with the reference's defaults (heracles/result.py:53-72):
    ell = arange, lower = ell, upper = lower shifted by one bin, weight = ones; one array shared by all axes unless a tuple."""
    given = getattr(result, name, None)
    if given is None:
        shape = np.shape(result)
        if name == "ell":
            return tuple(np.arange(shape[a]) for a in axes)
        if name == "lower":
            return _axis_arrays(result, "ell", axes)
        if name == "upper":
            return tuple(np.append(lo[1:], lo[-1] + 1) for lo in _axis_arrays(result, "lower", axes))
        if name == "weight":
            return tuple(np.ones(shape[a]) for a in axes)
        raise ValueError(f"cannot make default for array {name!r}")
    return given if isinstance(given, tuple) else (given,) * len(axes)


def result_axes(result):
    """Normalised tuple of the angular axes of a result (a bare array: its last axis)."""
    nd = np.ndim(result)
    axis = getattr(result, "axis", None)
    if axis is None:
        ell = getattr(result, "ell", None)
        axis = () if nd == 0 else tuple(range(nd - len(ell), nd)) if isinstance(ell, tuple) else (nd - 1,)
    elif isinstance(axis, (int, np.integer)):
        axis = (int(axis),)
    out = []
    for a in axis:
        if not -nd <= a < nd:
            raise np.exceptions.AxisError(a, nd) if hasattr(np, "exceptions") else ValueError(f"axis {a} out of bounds")
        out.append(a % nd)
    if len(set(out)) != len(out):
        raise ValueError("repeated axis")
    return tuple(out)


class BinPlan:
    """The binning of ONE angular axis: ``which[l]`` = bin of multipole position l (-1: none), ``w[l]`` its weight, ``norm[b]`` the
    summed weight of bin b, ``ell[b]`` the weighted mean multipole, ``lower`` / ``upper`` the edges."""

    def __init__(self, ell, edges, weight=None, result_weight=None):
        ell = np.asarray(ell)
        edges = np.asarray(edges)
        if edges.ndim != 1 or edges.size < 1:
            raise ValueError("bins must be a one-dimensional array of edges")
        base = np.ones(ell.shape) if result_weight is None else np.asarray(result_weight)
        if weight is None:
            w = base
        elif isinstance(weight, str):
            if weight not in _WEIGHT_RULES:
                raise ValueError(f"unknown weights string: {weight}")
            w = _WEIGHT_RULES[weight](ell) * base
        else:
            w = np.asarray(weight)[: base.size] * base
        nb = edges.size - 1
        slot = np.digitize(ell, edges)  # 0: below the first edge, edges.size: from the last edge on
        self.nbins = nb
        self.which = np.where((slot >= 1) & (slot <= nb), slot - 1, -1).astype(np.int32)
        self.w = np.asarray(w, dtype=float)
        inside = self.which >= 0
        self.norm = np.bincount(self.which[inside], weights=self.w[inside], minlength=nb)[:nb] if nb else np.zeros(0)
        mean = np.bincount(self.which[inside], weights=(self.w * ell)[inside], minlength=nb)[:nb] if nb else np.zeros(0)
        self.ell = _ratio(mean, self.norm)
        self.lower, self.upper = edges[:-1], edges[1:]

    def operator(self):
        """(nbins, n) array B with B[b, l] = w_l for l in bin b: numerators = B @ x."""
        op = np.zeros((self.nbins, self.which.size))
        inside = np.flatnonzero(self.which >= 0)
        op[self.which[inside], inside] = self.w[inside]
        return op

    def apply(self, array, axis):
        """Bin ``array`` along ``axis``."""
        moved = np.moveaxis(np.asarray(array, dtype=float), axis, -1)
        num = moved @ self.operator().T
        return np.moveaxis(_ratio(num, self.norm), -1, axis)


def plans_for(result, bins, weight=None):
    """One ``BinPlan`` per angular axis of ``result`` (bins / weight: one for all axes, or a tuple with one entry per axis)."""
    axes = result_axes(result)
    bins = bins if isinstance(bins, tuple) else (bins,) * len(axes)
    if len(bins) != len(axes):
        raise ValueError("result and bins have different number of ell axes")
    weight = weight if isinstance(weight, tuple) else (weight,) * len(axes)
    if len(weight) != len(axes):
        raise ValueError("result and weight have different number of ell axes")
    ells = _axis_arrays(result, "ell", axes)
    have = _axis_arrays(result, "weight", axes)
    return axes, [BinPlan(e, b, w, h) for e, b, w, h in zip(ells, bins, weight, have)]


def wrap_binned(array, spin, axes, plans, metadata=None):
    """The ``Result`` of a binned array: float dtype carrying ``metadata``, angular arrays from the plans (plain arrays for one
    axis, tuples for several: heracles/result.py:230-248)."""
    dt = np.dtype(float, metadata=dict(metadata or {}))
    array = np.ascontiguousarray(array, dtype=float).view(dt)
    parts = {name: tuple(getattr(p, attr) for p in plans)
             for name, attr in (("ell", "ell"), ("lower", "lower"), ("upper", "upper"), ("weight", "norm"))}
    if len(plans) == 1:
        parts = {name: value[0] for name, value in parts.items()}
    return Result(array, spin=spin, axis=axes, **parts)


def binned(result, bins, weight=None):
    """``heracles.result.binned``: bin a result (or every result of a mapping) along its angular axes."""
    if isinstance(result, Mapping):
        return {key: binned(value, bins, weight) for key, value in result.items()}
    axes, plans = plans_for(result, bins, weight)
    out = np.array(result, dtype=float)
    for axis, plan in zip(axes, plans):
        out = plan.apply(out, axis)
    md = getattr(getattr(result, "dtype", None), "metadata", None)
    return wrap_binned(out, getattr(result, "spin", None), axes, plans, md)
