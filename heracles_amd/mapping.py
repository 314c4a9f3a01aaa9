"""``heracles.transform`` (heracles/mapping.py:113-175) as one batched call per mapper.

The reference walks ``data`` -- ``{(field name, bin): map}`` -- and transforms one map at a time (mapping.py:171).  Here the walk
only *plans*: every map becomes a ``_Job`` filed under the mapper that will transform it, and each mapper then receives all of its
maps in a single ``transform_many`` call (``hx_map2alm_list``: one upload pipeline, no stacked host copy).  What a caller of the
reference can observe is kept: the keys of ``out`` in the order of ``data``, a map without ``spin`` metadata taking its field's, and
the two ``ValueError`` messages (unknown field name, spin mismatch), which are the reference's."""

from dataclasses import dataclass
from typing import Any

from .core import TocDict, update_metadata

__all__ = ["transform"]


@dataclass
class _Job:
    key: tuple
    array: Any
    spin: int
    alm: Any = None


def _declared_spin(array):
    """``spin`` from the dtype metadata of a numpy map; device tensors carry none."""
    meta = getattr(getattr(array, "dtype", None), "metadata", None)
    return None if not meta else meta.get("spin")


def _plan(fields, data, progress):
    """One pass over ``data``: validate every map against its field and file it under its mapper (first-seen order of mappers)."""
    jobs, per_mapper = [], {}
    for n, (key, entry) in enumerate(data.items(), 1):
        if progress is not None:
            progress.update(n, len(data))
        name = key[0]
        if name not in fields:
            raise ValueError(f"unknown field name: {name}")
        field = fields[name]
        array = getattr(entry, "array", entry)
        declared = _declared_spin(array)
        if declared is None:
            if hasattr(getattr(array, "dtype", None), "metadata"):
                update_metadata(array, spin=field.spin)
        elif declared != field.spin:
            raise ValueError(f"spin mismatch for field {name!r}: map has spin {declared}, field has spin {field.spin}")
        job = _Job(key, array, field.spin)
        jobs.append(job)
        mapper = getattr(field, "mapper_or_error", None) or field.mapper
        per_mapper.setdefault(id(mapper), (mapper, []))[1].append(job)
    return jobs, list(per_mapper.values())


def transform(fields, data, *, out=None, progress=None, device=None):
    """Alms of the maps in ``data`` for the ``fields`` they belong to; ``out`` (any mutable mapping) receives ``out[k, i]`` in the
    order of ``data``.  ``device="cuda"`` (not in the reference): the alms stay in HBM as ``DeviceArray``s, which
    ``angular_power_spectra`` takes as they are."""
    if out is None:
        out = TocDict()
    jobs, batches = _plan(fields, data, progress)
    for mapper, batch in batches:
        if hasattr(mapper, "transform_many"):
            extra = {} if device is None else {"device": device}
            alms = mapper.transform_many([j.array for j in batch], [j.spin for j in batch], **extra)
        else:  # a mapper of the reference: map by map, as it does
            alms = [mapper.transform(j.array, spin=j.spin) for j in batch]
        for job, alm in zip(batch, alms):
            job.alm = alm
    for job in jobs:
        out[job.key] = job.alm
    return out
