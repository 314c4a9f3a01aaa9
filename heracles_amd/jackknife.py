"""Delete-1 / delete-2 jackknife spectra with every alm resident in HBM.

The reference (heracles/dices/jackknife.py:41-248) transforms the maps of each of the njk + 1 region selections,
writes the alms to FITS files, and for every region combination reads them back, subtracts the deleted regions'
alms from the full-footprint alms and runs the all-pairs Cl sweep again -- the heaviest repeat caller of the
transform / two-point path.  Here the region maps are cut on the device (``hx_region_maps``), the transforms of all
(field, bin) maps of a region are batched per spin, the (njk + 1) alm sets stay in HBM, the delete-k alms are
formed there (``hx_alm_subtract``) and handed to the all-pairs kernel as device pointers; only the Cl blocks
(O(lmax) each) reach the host, where the bias and footprint corrections of the reference are applied.

Memory: (njk + 1) x ncomp x nlm x 16 bytes (nside 1024 / lmax 1500, 30 components, 40 regions: 22 GB of the 288).
"""

from __future__ import annotations

from dataclasses import replace
from itertools import combinations

import numpy as np

from . import _lib, sht
from .core import DeviceArray, update_metadata
from .twopoint import angular_power_spectra


class RegionAlms:
    """alms of every (field, bin) map for every region selection k = 0 (full maps) ... njk, on the device.

    ``keys`` are the map keys in transform order, ``meta[key]`` the dtype-metadata the mapper would attach,
    ``comps[key]`` the component slice of the key in ``tensor`` (njk + 1, ncomp, nlm)."""

    def __init__(self, keys, meta, comps, tensor):
        self.keys, self.meta, self.comps, self.tensor = keys, meta, comps, tensor

    @property
    def njk(self):
        return self.tensor.shape[0] - 1

    def _dress(self, comp_tensor):
        out = {}
        for key in self.keys:
            sl = self.comps[key]
            t = comp_tensor[sl]
            out[key] = DeviceArray(t[0] if self.meta[key].get("spin", 0) == 0 else t, self.meta[key])
        return out

    def full(self):
        """{key: DeviceArray} of the full-footprint alms."""
        return self._dress(self.tensor[0])

    def delete(self, regions, work=None):
        """{key: DeviceArray} of the alms with the given regions removed: full - sum of the regions' alms
        (heracles/dices/jackknife.py:222-225, :298-304), formed on the device into ``work`` (ncomp, nlm)."""
        import ctypes
        import torch

        ncomp, nlm = self.tensor.shape[1:]
        if work is None:
            work = torch.empty((ncomp, nlm), dtype=torch.complex128, device=self.tensor.device)
        subs = [self.tensor[r] for r in regions]
        ptrs = (ctypes.c_void_p * max(len(subs), 1))(*[t.data_ptr() for t in subs])
        _lib.check(_lib.load().hx_alm_subtract(ncomp * nlm, _lib.ptr(self.tensor[0]), len(subs), ptrs, _lib.ptr(work)))
        return self._dress(work)


def _mapper_of(field):
    m = getattr(field, "mapper_or_error", None)
    return m if m is not None else field.mapper


def region_alms(fields, maps, jk_map, *, device="cuda"):
    """Transforms of ``maps`` (``{(field key, bin): map}``) restricted to every jackknife region, batched per spin
    and kept in HBM: the counterpart of heracles.dices.jackknife.compute_jk_alms (jackknife.py:93-128, :143-148).
    All fields must share one HEALPix resolution and band limit (one plan); their mappers supply the weights,
    iterations and pixel-window deconvolution exactly as ``HipHealpixMapper.transform`` would."""
    import torch

    _lib.ensure_init()
    keys = list(maps)
    if not keys:
        raise ValueError("no maps")
    jk_map = np.asarray(jk_map)
    njk = len(np.unique(jk_map)[np.unique(jk_map) != 0])
    mappers = {k: _mapper_of(fields[k[0]]) for k in keys}
    m0 = mappers[keys[0]]
    if any((m.nside, m.lmax) != (m0.nside, m0.lmax) for m in mappers.values()):
        raise NotImplementedError("region_alms: all fields must share nside and lmax")
    plan = sht.get_plan(m0.nside, m0.lmax)
    npix, nlm = plan.npix, plan.nlm
    by_spin = {0: [k for k in keys if fields[k[0]].spin == 0], 2: [k for k in keys if fields[k[0]].spin == 2]}
    if len(by_spin[0]) + len(by_spin[2]) != len(keys):
        raise NotImplementedError("spin-0 and spin-2 fields only")
    order = by_spin[0] + by_spin[2]
    comps, meta, c = {}, {}, 0
    for k in order:
        n = 2 if fields[k[0]].spin else 1
        comps[k] = slice(c, c + n)
        c += n
        md = dict(np.asarray(maps[k]).dtype.metadata or {})
        md.setdefault("spin", fields[k[0]].spin)
        md["deconv"] = mappers[k].deconvolve
        meta[k] = md
    ncomp = c
    out = torch.empty((njk + 1, ncomp, nlm), dtype=torch.complex128, device=device)
    region = torch.as_tensor(np.ascontiguousarray(jk_map, dtype=np.float64)).to(device)
    L = _lib.load()

    # the weight file of a configured data path is part of a mapper's settings: load it before the mappers are compared, as
    # HipHealpixMapper.transform does -- the reference transforms every jackknife map through mapper.transform
    # (dices/jackknife.py:143-148), so the two must weight alike whichever runs first
    for mp in {id(m): m for m in mappers.values()}.values():
        if hasattr(mp, "_load_weights"):
            mp._load_weights()

    def settings(k):
        """What a mapper contributes to the transform besides (nside, lmax): the fields of one batched call must agree on all
        of it -- the reference transforms every field with its OWN mapper (dices/jackknife.py:143-148 via mapping.transform)."""
        mp = mappers[k]
        pw = getattr(mp, "pixwin", None)
        return (fields[k[0]].spin, bool(mp.deconvolve), id(pw) if pw is not None else None, id(mp.ring_weights) if mp.ring_weights is not None else None,
                id(mp.pixel_weights) if mp.pixel_weights is not None else None, int(mp.niter))

    groups = {}
    for k in order:
        groups.setdefault(settings(k), []).append(k)
    for sett, ks in groups.items():
        spin = sett[0]
        mp = mappers[ks[0]]
        fl = mp._fl(spin)
        # the maps of the group go to one device tensor map by map through the library's staging pipeline (a stacked host copy
        # and torch's upload of pageable memory cost ~8 s for the 48 GB of the bench's job; this: ~1 s)
        parts = [np.ascontiguousarray(np.asarray(maps[k], dtype=np.float64)).reshape(-1, npix) for k in ks]
        dmaps = torch.empty((sum(p.shape[0] for p in parts), npix), dtype=torch.float64, device=device)
        r = 0
        for p in parts:
            _lib.copy(dmaps[r : r + p.shape[0]], p)
            r += p.shape[0]
        del parts
        scratch = torch.empty_like(dmaps)
        # the keys of a group need not be adjacent in `order`: transform into a work tensor, scatter per key
        contiguous = all(comps[ks[i + 1]].start == comps[ks[i]].stop for i in range(len(ks) - 1))
        work = None if contiguous else torch.empty((dmaps.shape[0], nlm), dtype=torch.complex128, device=device)
        c0 = comps[ks[0]].start
        for k in range(njk + 1):
            src = dmaps
            if k > 0:
                _lib.check(L.hx_region_maps(npix, dmaps.shape[0], _lib.ptr(dmaps), _lib.ptr(region), float(k), _lib.ptr(scratch)))
                src = scratch
            dst = out[k, c0 : c0 + dmaps.shape[0]] if contiguous else work
            plan.map2alm(src, spin, ring_weights=mp.ring_weights, pix_weights=mp.pixel_weights, fl=fl, niter=mp.niter, out=dst)
            if not contiguous:
                # scatter on the LIBRARY's stream (hx_copy, device to device): the next iteration's hx_region_maps / map2alm(out=work)
                # run on that stream too and are ordered behind these copies -- a torch-stream copy would leave the reuse of `work` unordered
                r = 0
                for kk in ks:
                    n = comps[kk].stop - comps[kk].start
                    _lib.copy(out[k, comps[kk]], work[r : r + n])
                    r += n
        del dmaps, scratch, work
    return RegionAlms(order, meta, comps, out)


# ---- host-side corrections of the reference, restated (O(lmax) per spectrum) ---------------------------------------
def jackknife_fsky(jk_map, jk=0, jk2=0, ratio=True):
    """Sky fraction left after removing regions jk and jk2, relative to the footprint (pixels with a non-zero
    region label) if ``ratio`` (heracles/dices/jackknife.py:349-367)."""
    jk_map = np.asarray(jk_map)
    inside = jk_map > 0
    kept = np.count_nonzero(inside & (jk_map != jk) & (jk_map != jk2)) / jk_map.size
    return kept / (np.count_nonzero(inside) / jk_map.size) if ratio else kept


def correct_bias(cls, jk_map, jk=0, jk2=0):
    """Put the full-footprint bias back and take the bias of the reduced footprint out (bias scales with the sky
    fraction kept; heracles/dices/jackknife.py:389-416): the reference adds and subtracts the bias as a scalar over
    the whole block, which is kept; the ``bias`` metadata is updated."""
    f = jackknife_fsky(jk_map, jk=jk, jk2=jk2)
    out = {}
    for key, res in cls.items():
        b = (res.array.dtype.metadata or {}).get("bias", 0)
        md = dict(res.array.dtype.metadata or {})
        arr = np.array(res.array) + b - b * f
        update_metadata(arr, **{**md, "bias": b * f})
        out[key] = replace(res, array=arr)
    return out


def correct_footprint_fsky(cls, jk_map, jk=0, jk2=0, unmixed=False):
    """"Fast" footprint correction: divide by the sky fraction kept (relative to the footprint unless the spectra
    are unmixed; heracles/dices/jackknife.py:419-437)."""
    f = jackknife_fsky(jk_map, jk=jk, jk2=jk2, ratio=not unmixed)
    return {key: replace(res, array=res.array / f) for key, res in cls.items()}


def correct_footprint_naturalspice(cls, cls_mm, mls0, fields, unmixed=False):
    """"Full" footprint correction through the mask correlation functions (heracles/dices/jackknife.py:440-470):
    alpha = xi(jackknife mask) [/ xi(full mask)], data correlation divided by alpha, back to Cl."""
    from .transforms import cl2corr, corr2cl
    from .unmixing import _naturalspice, _pad

    w0, wjk = cl2corr(mls0), cl2corr(cls_mm)
    alphas = {}
    for key in wjk:
        alpha = wjk[key].array
        if not unmixed:
            # the reference divides without a guard (dices/jackknife.py:420): columns a mask pair does not populate (e.g. the B-mode
            # columns of a scalar mask) are 0 / 0 = nan there too, and _naturalspice never reads them -- same values, no warning
            with np.errstate(invalid="ignore", divide="ignore"):
                alpha = alpha / w0[key].array
        alphas[key] = replace(mls0[key], array=alpha)
    first_cls, first_mls = next(iter(cls.values())), next(iter(mls0.values()))
    lmax = first_cls.shape[first_cls.axis[0]]
    lmax_mask = first_mls.shape[first_mls.axis[0]]
    wcls = _naturalspice(cl2corr(_pad(cls, lmax_mask)), alphas, fields)
    return _pad(corr2cl(wcls), lmax)


def jackknife_cls(data_maps, vis_maps, jk_map, fields, mask_correction="Fast", unmixed=False, nd=1, progress=None,
                  device="cuda"):
    """Spectra of the delete-``nd`` jackknife samples: ``{regions: {(f1, f2, i1, i2): Result}}`` as
    heracles.dices.jackknife.jackknife_cls (jackknife.py:41-90) returns them (nd = 0: ``{(): cls of the full maps}``),
    without the FITS round trips: region alms, delete-k alms and the all-pairs sweeps stay on the device."""
    if nd not in (0, 1, 2):
        raise ValueError("number of deletions must be 0, 1 or 2")
    if mask_correction not in ("Fast", "Full"):
        raise ValueError("mask_correction must be 'Fast' or 'Full'")
    data = region_alms(fields, data_maps, jk_map, device=device)
    if nd == 0:
        return {(): angular_power_spectra(data.full())}
    vis = mls0 = None
    if mask_correction == "Full":
        vis = region_alms(fields, vis_maps, jk_map, device=device)
        mls0 = angular_power_spectra(vis.full())
    njk = data.njk
    combos = list(combinations(range(1, njk + 1), nd))
    out = {}
    work = work_v = None
    import torch

    work = torch.empty(tuple(data.tensor.shape[1:]), dtype=torch.complex128, device=data.tensor.device)
    if vis is not None:
        work_v = torch.empty(tuple(vis.tensor.shape[1:]), dtype=torch.complex128, device=vis.tensor.device)
    for n, regions in enumerate(combos):
        if progress is not None:
            progress.update(n, len(combos))
        cls = angular_power_spectra(data.delete(regions, work))
        cls = correct_bias(cls, jk_map, *regions)
        if mask_correction == "Full":
            cls_mm = angular_power_spectra(vis.delete(regions, work_v))
            cls = correct_footprint_naturalspice(cls, cls_mm, mls0, fields, unmixed=unmixed)
        else:
            cls = correct_footprint_fsky(cls, jk_map, *regions, unmixed=unmixed)
        out[regions] = cls
    if progress is not None:
        progress.update(len(combos), len(combos))
    return out
