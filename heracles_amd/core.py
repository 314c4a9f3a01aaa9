"""Small host-side helpers mirroring the parts of heracles.core / heracles.result that the
hot path touches: dtype-metadata pass-through (heracles/core.py:102-122), the pattern
matching of two-point keys (core.py:34-58) and a minimal Result container
(heracles/result.py:75-121).  If the user's own ``heracles`` is importable its classes
are used instead, so objects returned from here are the types Heracles expects.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any

import numpy as np

try:  # (the branch an installed Heracles takes; run by tests/test_production_config.py through the reference tree)
    from heracles.core import TocDict as _TocDict, toc_match as _toc_match
    from heracles.core import update_metadata as _update_metadata
    from heracles.result import Result as _Result

    HAVE_HERACLES = True
except Exception:  # noqa: BLE001 - any import problem means "not available"
    HAVE_HERACLES = False


def update_metadata(array, **metadata):
    """Attach/merge metadata on ``array.dtype`` in place (same contract as core.py:102)."""
    md = dict(array.dtype.metadata or {})
    md.update(metadata)
    dt = array.dtype
    base = dt.fields if dt.fields is not None else dt.str
    new = np.dtype(base, metadata=md)
    if not np.can_cast(new, array.dtype, casting="no"):
        raise ValueError("array with unsupported dtype")
    array.dtype = new


def toc_match(key, include=None, exclude=None) -> bool:
    """Whether a key passes include / exclude pattern lists; ``...`` is a wildcard."""
    if not isinstance(key, tuple):
        key = (key,)

    def hit(pattern):
        return all(p is Ellipsis or p == k for p, k in zip(pattern, key))

    if include is not None and not any(hit(p) for p in include):
        return False
    if exclude is not None and any(hit(p) for p in exclude):
        return False
    return True


def _prefix_matches(pattern, key):
    """A pattern is a key prefix whose ``...`` entries match anything; a non-tuple key is a one-element key."""
    parts = key if isinstance(key, tuple) else (key,)
    if len(pattern) > len(parts) or (not isinstance(key, tuple) and len(pattern) != 1):
        return False
    return all(want is ... or want == have for want, have in zip(pattern, parts))


class TocDict(dict):
    """dict with the selection rule of heracles.core.TocDict (heracles/core.py:63-99): an exact key returns its value;
    anything else is read as a prefix pattern and returns the sub-dictionary of matching keys (KeyError if empty)."""

    def __getitem__(self, pattern):
        try:
            if dict.__contains__(self, pattern):
                return dict.__getitem__(self, pattern)
        except TypeError:  # unhashable pattern: can only be a selection
            pass
        prefix = pattern if isinstance(pattern, tuple) else (pattern,)
        if not prefix:  # the empty pattern selects everything
            return TocDict(self)
        selected = TocDict((k, v) for k, v in self.items() if _prefix_matches(prefix, k))
        if not selected:
            raise KeyError(prefix)
        return selected


def _axis_tuple(axis, ndim, ell):
    if axis is None:
        if ndim == 0:
            return ()
        if isinstance(ell, tuple):
            return tuple(range(ndim - len(ell), ndim))
        return (ndim - 1,)
    if isinstance(axis, int):
        axis = (axis,)
    return tuple(a % ndim for a in axis)


@dataclass(frozen=True, repr=False)
class Result:
    """Array plus angular axes, as heracles.result.Result."""

    array: Any
    ell: Any = None
    spin: Any = None
    axis: Any = None
    lower: Any = None
    upper: Any = None
    weight: Any = None

    def __post_init__(self):
        object.__setattr__(self, "axis", _axis_tuple(self.axis, np.ndim(self.array), self.ell))

    def __repr__(self):
        return f"Result(axis={self.axis!r})"

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.array, dtype=dtype)

    def __getitem__(self, key):
        return self.array[key]

    @property
    def ndim(self):
        return self.array.ndim

    @property
    def shape(self):
        return self.array.shape

    @property
    def dtype(self):
        return self.array.dtype


if HAVE_HERACLES:
    TocDict = _TocDict  # noqa: F811
    toc_match = _toc_match  # noqa: F811
    update_metadata = _update_metadata  # noqa: F811
    Result = _Result  # noqa: F811


class DeviceArray:
    """A device-resident array (a contiguous torch CUDA tensor) dressed like the numpy arrays the two-point
    drivers receive: ``shape`` and ``dtype`` (with the Heracles metadata dict attached, which a torch tensor
    cannot carry).  ``angular_power_spectra`` hands its rows to the all-pairs kernel as device pointers."""

    def __init__(self, tensor, metadata=None):
        self.tensor = tensor
        self.shape = tuple(tensor.shape)
        kind = np.complex128 if tensor.is_complex() else np.float64
        self.dtype = np.dtype(kind, metadata=dict(metadata or {}))

    @property
    def ndim(self):
        return len(self.shape)

    def rows(self):
        """1-D views, one per component."""
        flat = self.tensor.reshape(-1, self.shape[-1])
        return [flat[k] for k in range(flat.shape[0])]

    def numpy(self):
        out = self.tensor.cpu().numpy()
        update_metadata(out, **(self.dtype.metadata or {}))
        return out
