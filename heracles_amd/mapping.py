"""``heracles.transform`` (heracles/mapping.py:113-175) as a batched call.

The reference walks ``data`` -- ``{(field name, bin): map}`` -- and calls ``field.mapper_or_error.transform(m, spin=s)`` once per
map (mapping.py:171).  This mirror keeps its interface, key order, metadata rule and errors (unknown field name, spin mismatch:
``ValueError`` with the reference's messages) and hands the maps of every mapper to ``transform_many`` in one call
(``hx_map2alm_list``: one upload pipeline for all of them, no stacked host copy); a mapper without ``transform_many`` is called map
by map, as in the reference."""

from .core import TocDict, update_metadata

__all__ = ["transform"]


def _mapper_of(field):
    m = getattr(field, "mapper_or_error", None)
    return m if m is not None else field.mapper


def transform(fields, data, *, out=None, progress=None, device=None):
    """Alms of the maps in ``data`` for the ``fields`` they belong to; ``out`` (any mutable mapping) receives ``out[k, i]`` in the
    order of ``data``.  ``device="cuda"`` (not in the reference): the alms stay in HBM as ``DeviceArray``s, which
    ``angular_power_spectra`` takes as they are."""
    if out is None:
        out = TocDict()
    items = []
    current, total = 0, len(data)
    for (k, i), m in data.items():
        current += 1
        if progress is not None:
            progress.update(current, total)
        m = getattr(m, "array", m)
        try:
            field = fields[k]
        except KeyError:
            msg = f"unknown field name: {k}"
            raise ValueError(msg) from None
        s = field.spin
        m_spin = (m.dtype.metadata or {}).get("spin")
        if m_spin is None:
            update_metadata(m, spin=s)
        elif m_spin != s:
            msg = f"spin mismatch for field {k!r}: map has spin {m_spin}, field has spin {s}"
            raise ValueError(msg)
        items.append(((k, i), m, s, _mapper_of(field)))
    groups = {}
    for it in items:
        groups.setdefault(id(it[3]), []).append(it)
    alms = {}
    for group in groups.values():
        mapper = group[0][3]
        if hasattr(mapper, "transform_many"):
            kw = {} if device is None else {"device": device}
            res = mapper.transform_many([it[1] for it in group], [it[2] for it in group], **kw)
        else:
            res = [mapper.transform(it[1], spin=it[2]) for it in group]
        for it, a in zip(group, res):
            alms[it[0]] = a
    for key, _, _, _ in items:
        out[key] = alms[key]
    return out
