"""Two-point functions of the hot path, backed by libhxsht.

Mirrors the call surface of ``heracles.twopoint`` (heracles/twopoint.py): ``alm2lmax``,
``alm2cl``, ``angular_power_spectra``, ``debias_cls``, ``mixing_matrices`` and the
``convolvecl.mixmat`` / ``mixmat_eb`` functions the reference imports at
heracles/twopoint.py:330.
"""

from __future__ import annotations

import ctypes as C
import logging
import time
from itertools import combinations_with_replacement, product

import numpy as np

from . import _lib
from .binning import BinPlan, binned, wrap_binned
from .core import DeviceArray, Result, TocDict, toc_match, update_metadata

logger = logging.getLogger(__name__)


def alm2lmax(alm, mmax=None):
    """lmax of an alm array from its last-axis length (heracles/twopoint.py:55-60)."""
    n = np.shape(alm)[-1]
    return (int((8 * n + 1) ** 0.5 + 0.01) - 3) // 2


def _is_tensor(a):
    return hasattr(a, "data_ptr") and not isinstance(a, np.ndarray)


def alm2cl_pairs(comps, pairs, lmax_out, m_range=None, out=None):
    """All requested component-pair spectra in one launch.

    comps: list of 1-D complex128 arrays (numpy, or torch CUDA tensors); pairs: list of
    (i, j) indices into comps.  Returns a float64 array (npairs, lmax_out+1).
    m_range = (m0, m1[, step]): the sum over the orders m0, m0 + step, ... < m1 only (a rank's share on the m-sharded route).
    out: the destination (C-contiguous float64 of that shape; a loop passes the same page-locked array -- ``pinned_empty`` -- every
    time and saves the first touch and the staged copy of a fresh 23 MB result: 3 ms per call at lmax 6144 / 475 spectra).
    """
    L = _lib.load()
    _lib.ensure_init()
    ncomp = len(comps)
    keep = []
    ptrs = (C.c_void_p * ncomp)()
    lmaxs = (C.c_int * ncomp)()
    for k, a in enumerate(comps):
        if not _is_tensor(a):
            a = np.ascontiguousarray(a, dtype=np.complex128)
        elif not a.is_contiguous():
            a = a.contiguous()
        keep.append(a)
        ptrs[k] = _lib.ptr(a).value
        lmaxs[k] = alm2lmax(a)
    npairs = len(pairs)
    pi = (C.c_int * max(npairs, 1))(*[p[0] for p in pairs])
    pj = (C.c_int * max(npairs, 1))(*[p[1] for p in pairs])
    out = _lib.result_array((npairs, lmax_out + 1), out)  # (every element is written by the call)
    if npairs and m_range is None:
        _lib.check(L.hx_alm2cl_pairs(ncomp, lmaxs, ptrs, int(lmax_out), npairs, pi, pj, _lib.ptr(out)))
    elif npairs:
        step = int(m_range[2]) if len(m_range) > 2 else 1
        _lib.check(L.hx_alm2cl_pairs_range(ncomp, lmaxs, ptrs, int(lmax_out), npairs, pi, pj, int(m_range[0]), int(m_range[1]), step, _lib.ptr(out)))
    return out


def alm2cl(alm, alm2=None, *, lmax=None):
    """Angular (cross-)power spectrum block, same semantics as heracles/twopoint.py:63-101.

    Leading axes are component axes and broadcast to a block
    ``(*alm.shape[:-1], *alm2.shape[:-1], L)``; the two inputs may have different lmax.
    """
    if alm2 is None:
        alm2 = alm
    alm = np.asanyarray(alm)
    alm2 = np.asanyarray(alm2)
    lmax1, lmax2 = alm2lmax(alm), alm2lmax(alm2)
    lout = min(lmax1, lmax2) if lmax is None else min(lmax, lmax1, lmax2)
    a = np.ascontiguousarray(alm, dtype=np.complex128).reshape(-1, alm.shape[-1])
    same = alm2 is alm
    b = a if same else np.ascontiguousarray(alm2, dtype=np.complex128).reshape(-1, alm2.shape[-1])
    comps = list(a) + ([] if same else list(b))
    offs = 0 if same else a.shape[0]
    pairs = [(i, offs + j) for i in range(a.shape[0]) for j in range(b.shape[0])]
    cl = alm2cl_pairs(comps, pairs, lout)
    return cl.reshape(*alm.shape[:-1], *alm2.shape[:-1], lout + 1)


def _pixwin_for(md, i, s, lmax, pixwin):
    """Pixel window of field i for debiasing (heracles/twopoint.py:147-165)."""
    if md.get(f"kernel_{i}") != "healpix":
        return None
    nside = md.get(f"nside_{i}")
    deconv = md.get(f"deconv_{i}", True)
    if nside is None or not deconv or s not in (0, 2):
        return None
    from .mapper import pixel_window

    pw0, pw2 = pixel_window(nside, lmax, pixwin)
    return pw0 if s == 0 else pw2


def _auto_bias(md, both_spin2):
    """Additive bias of an auto-spectrum from the catalogue statistics the field attached to its maps
    (fsky * <mu^2> / density, halved per E/B component for a spin-2 field; heracles/twopoint.py:260-269);
    None when any of the three is absent."""
    try:
        product_ = md["fsky"] * md["musq"] / md["dens"]
    except (KeyError, TypeError):
        return None
    if md["fsky"] is None or md["musq"] is None or md["dens"] is None:
        return None
    return 0.5 * product_ if both_spin2 else product_


def _bias_profile(shape, md, bias, pixwin):
    """The additive term of one spectrum block as an array of the block's shape: `bias` from l = max|spin| on (below
    that the harmonic space of the field is empty), on the EE and BB diagonal only for a spin-2 x spin-2 block, and
    divided by the pixel window of every HEALPix side whose window was deconvolved (heracles/twopoint.py:126-165)."""
    spins = (md.get("spin_1", 0), md.get("spin_2", 0))
    first = max(abs(sp) for sp in spins)
    nl = shape[-1]
    profile = np.zeros(nl)
    profile[first:] = bias
    for side, sp in enumerate(spins, start=1):
        window = _pixwin_for(md, side, sp, nl - 1, pixwin)
        if window is not None:
            profile[first:] /= window[first:]
    term = np.zeros(shape)
    if all(spins):
        if tuple(shape[:2]) != (2, 2):
            raise AssertionError("a spin-2 x spin-2 block must have shape (2, 2, ...)")
        term[0, 0] = profile
        term[1, 1] = profile
    else:
        term[...] = profile
    return term


def _debias_cl(cl, bias=None, md=None, *, inplace=False, pixwin=None):
    """Remove additive bias from a spectrum block (heracles/twopoint.py:104-170)."""
    meta = (cl.dtype.metadata or {}) if md is None else md
    target = cl
    if not inplace:
        target = cl.copy()
        update_metadata(target, **meta)
    amount = meta.get("bias") if bias is None else bias
    if amount is not None:
        target[:] -= _bias_profile(target.shape, meta, amount, pixwin)
    return target


def debias_cls(cls, bias=None, *, inplace=False, pixwin=None):
    """Remove bias from a set of spectra (heracles/twopoint.py:302-313)."""
    out = cls if inplace else TocDict()
    for key in cls:
        out[key] = _debias_cl(cls[key], bias and bias.get(key), inplace=inplace, pixwin=pixwin)
    return out


def angular_power_spectra(alms, alms2=None, *, lmax=None, debias=True, bins=None, weights=None,
                          include=None, exclude=None, out=None, pixwin=None):
    """All auto/cross spectra of a set of alms: heracles/twopoint.py:173-299.

    Pair enumeration, duplicate skipping, key canonicalisation, metadata merge and the
    auto-spectrum bias rule follow the reference; the arithmetic of every pair is done by
    ONE all-pairs launch (each alm is read from HBM once per component tile instead of
    once per partner).
    """
    logger.info("computing cls for %d%s alm(s)", len(alms), f"x{len(alms2)}" if alms2 is not None else "")
    t0 = time.monotonic()
    if alms2 is None:
        pairs = combinations_with_replacement(alms, 2)
        alms2 = alms
    else:
        pairs = product(alms, alms2)
    names = set()
    cls = TocDict() if out is None else out
    # pass 1: host logic exactly as the reference's loop, collecting the work list
    todo = []
    seen = set()
    for (k1, i1), (k2, i2) in pairs:
        if (k1, k2, i1, i2) in cls or (k2, k1, i2, i1) in cls:
            continue
        if (k1, k2, i1, i2) in seen or (k2, k1, i2, i1) in seen:
            continue
        if (k1, k2) not in names and (k2, k1) in names:
            i1, i2 = i2, i1
            k1, k2 = k2, k1
            swapped = True
        else:
            swapped = False
        if not toc_match((k1, k2, i1, i2), include, exclude):
            continue
        if swapped:
            alm1, alm2 = alms2[k1, i1], alms[k2, i2]
        else:
            alm1, alm2 = alms[k1, i1], alms2[k2, i2]
        todo.append(((k1, k2, i1, i2), alm1, alm2))
        seen.add((k1, k2, i1, i2))
        names.add((k1, k2))
    # pass 2: one launch for every component pair of every map pair
    louts = []
    for key, alm1, alm2 in todo:
        l1, l2 = alm2lmax(alm1), alm2lmax(alm2)
        louts.append(min(l1, l2) if lmax is None else min(lmax, l1, l2))
    # the kernel writes a common output length; group by output lmax.  Each group stages only the
    # components its own pairs use (alms of different band limits may be mixed, twopoint.py:78-99).
    results = [None] * len(todo)
    for lo in sorted(set(louts)):
        comps, index, plist, owners = [], {}, [], []

        def comp_ids(arr, comps=comps, index=index):
            if isinstance(arr, DeviceArray):
                a2 = arr.rows()  # device pointers: the alms never leave HBM
            else:
                a2 = np.ascontiguousarray(arr, dtype=np.complex128).reshape(-1, arr.shape[-1])
            ids = []
            for row in range(len(a2)):
                key = (id(arr), row)
                if key not in index:
                    index[key] = len(comps)
                    comps.append(a2[row])
                ids.append(index[key])
            return ids

        for n, (key, alm1, alm2) in enumerate(todo):
            if louts[n] != lo:
                continue
            ia, ib = comp_ids(alm1), comp_ids(alm2)
            owners.append((n, len(plist), len(ia), len(ib)))
            plist.extend((a, b) for a in ia for b in ib)
        block = alm2cl_pairs(comps, plist, lo)
        for n, start, na, nb in owners:
            key, alm1, alm2 = todo[n]
            results[n] = block[start : start + na * nb].reshape(*alm1.shape[:-1], *alm2.shape[:-1], lo + 1).copy()
    # pass 3: metadata, shot-noise bias and wrapping of every block (heracles/twopoint.py:247-287)
    for (key, alm1, alm2), cl in zip(todo, results):
        k1, k2, i1, i2 = key
        sides = (alm1.dtype.metadata or {}, alm2.dtype.metadata or {})
        if any("spin" not in side for side in sides):
            raise ValueError(f"missing spin metadata for {k1} or {k2}")
        s1, s2 = sides[0]["spin"], sides[1]["spin"]
        md = {f"{name}_{n}": value for n, side in enumerate(sides, start=1) for name, value in side.items()}
        if (k1, i1) == (k2, i2):
            bias = _auto_bias(sides[0], both_spin2=(s1 == 2 and s2 == 2))
            if bias is not None:
                md["bias"] = bias
                if debias:
                    _debias_cl(cl, bias, md, inplace=True, pixwin=pixwin)
        update_metadata(cl, **md)
        res = Result(cl, spin=(s1, s2), axis=-1)
        if bins is not None:
            res = binned(res, bins, weights)  # (O(lmax) per block: on the host, heracles/twopoint.py:283-284)
        cls[k1, k2, i1, i2] = res
    logger.info("computed %d cl(s) in %.3f s", len(cls), time.monotonic() - t0)
    return cls


# ---- mixing matrices -------------------------------------------------------------------
def _mm_args(cl, l1max, l2max, l3max):
    # Defaults of the third-party convolvecl functions are not in the reference tree
    # [UNVERIFIED]; chosen so that the shapes used by tests/test_twopoint.py:412-462 hold:
    # l3max = len(cl)-1, l1max = l3max, l2max = l1max.
    cl = np.ascontiguousarray(np.asarray(cl, dtype=np.float64))
    if cl.ndim != 1:
        raise ValueError("cl must be one-dimensional")
    if l3max is None:
        l3max = cl.shape[0] - 1
    if l1max is None:
        l1max = l3max
    if l2max is None:
        l2max = l1max
    return cl, int(l1max), int(l2max), int(l3max)


def mixmat(cl, l1max=None, l2max=None, l3max=None, spin=(0, 0), out=None):
    """Mixing matrix of a mask spectrum for spins (0,0), (0,2) or (2,0); replaces
    ``convolvecl.mixmat`` as called at heracles/twopoint.py:382-388.  Shape (l1max+1, l2max+1),
    axis 0 is the output multipole.  ``out``: an array the caller owns and re-uses (float64, that shape, C-contiguous; numpy --
    pageable or ``heracles_amd.pinned_empty`` -- or a torch tensor, host or device), filled and returned instead of a fresh one."""
    cl, l1max, l2max, l3max = _mm_args(cl, l1max, l2max, l3max)
    s1, s2 = spin
    out = _lib.result_array((l1max + 1, l2max + 1), out)
    _lib.ensure_init()
    _lib.check(_lib.load().hx_mixmat(_lib.ptr(cl), cl.shape[0], l1max, l2max, l3max, int(s1), int(s2), _lib.ptr(out)))
    return out


def mixmat_eb(cl, l1max=None, l2max=None, l3max=None, spin=(2, 2), out=None):
    """E/B mixing matrices (EE->EE, EE->BB, EB->EB); replaces ``convolvecl.mixmat_eb``.  ``out`` as in ``mixmat``, shape
    (3, l1max+1, l2max+1): a build into a fresh numpy array pays the first touch of its pages (0.9 GB at L = 6144: more than the GPU
    spends on the matrices); a loop that hands every result on passes the same ``out`` each time."""
    cl, l1max, l2max, l3max = _mm_args(cl, l1max, l2max, l3max)
    if tuple(abs(s) for s in spin) != (2, 2):
        raise NotImplementedError(f"mixmat_eb for spin {spin} not supported")
    out = _lib.result_array((3, l1max + 1, l2max + 1), out)
    _lib.ensure_init()
    _lib.check(_lib.load().hx_mixmat_eb(_lib.ptr(cl), cl.shape[0], l1max, l2max, l3max, _lib.ptr(out)))
    return out


def mixmat_release():
    """Free what ``mixmat`` / ``mixmat_eb`` keep in HBM between calls (the tables of the last (l1max, l2max, l3max) and the staging
    buffer of a host destination, ~3 GB at L = 6144: hx_mixmat_release)."""
    _lib.check(_lib.load().hx_mixmat_release())


class _NoProgress:
    def update(self, *a):
        pass

    def task(self, *a):
        from contextlib import nullcontext

        return nullcontext()


def mixing_requests(fields, cls):
    """What heracles.twopoint.mixing_matrices computes, as a list ((f1, f2, i1, i2), mask-cl key, (spin1, spin2)) in the
    order the reference produces it (heracles/twopoint.py:341-371): every field pair whose masks name a given mask
    spectrum, each unordered (field, bin) combination once."""
    users = {}
    for name, field in fields.items():
        if field.mask is not None:
            users.setdefault(field.mask, []).append(name)
    taken, todo = set(), []
    for ck in cls:
        m1, m2, i1, i2 = ck
        if m1 not in users or m2 not in users:
            continue
        for f1, f2 in product(users[m1], users[m2]):
            if (f1, f2, i1, i2) in taken or (f2, f1, i2, i1) in taken:
                continue
            taken.add((f1, f2, i1, i2))
            todo.append(((f1, f2, i1, i2), ck, (fields[f1].spin, fields[f2].spin)))
    return todo


class MixmatContext:
    """Mask-independent part of a mixing-matrix build (Gauss-Legendre nodes, Wigner-d tables, GEMM tiles) for one
    (l1max, l2max, l3max): built once on the GPU, then every mask spectrum costs its node weights and one GEMM per
    product (hx_mixctx_*)."""

    KIND = {"00": 1, "02": 2, "22": 4}

    def __init__(self, l1max, l2max, l3max):
        _lib.ensure_init()
        self.l1max, self.l2max, self.l3max = int(l1max), int(l2max), int(l3max)
        self._h = _lib.load().hx_mixctx_create(self.l1max, self.l2max, self.l3max)
        if not self._h:
            raise _lib.HxError(-1, _lib.load().hx_last_error().decode(errors="replace"))

    def result_buffer(self, spin=(2, 2)):
        """A page-locked host array of the shape ``self(cl, spin)`` returns, owned by this context (one per shape, freed with it):
        pass it as ``out=`` for every matrix of a loop that consumes each result before the next build -- the GPU writes it by DMA,
        no staging copy, no page faults.  The NEXT call with the same ``out`` overwrites it."""
        s1, s2 = (abs(int(v)) for v in spin)
        shape = (self.l1max + 1, self.l2max + 1)
        if (s1, s2) == (2, 2):
            shape = (3,) + shape
        if not hasattr(self, "_buffers"):
            self._buffers = {}
        if shape not in self._buffers:
            self._buffers[shape] = _lib.pinned_empty(shape)
        return self._buffers[shape]

    @staticmethod
    def _kind(spin):
        s1, s2 = (abs(int(v)) for v in spin)
        if (s1, s2) == (0, 0):
            return 1
        if sorted((s1, s2)) == [0, 2]:
            return 2
        if (s1, s2) == (2, 2):
            return 4
        raise NotImplementedError(f"mixing matrix for spin {tuple(spin)} not supported")

    def __call__(self, cl, spin, out=None):
        kind = self._kind(spin)
        cl = np.ascontiguousarray(np.asarray(cl, dtype=np.float64))
        shape = (self.l1max + 1, self.l2max + 1)
        out = _lib.result_array((3,) + shape if kind == 4 else shape, out)
        _lib.check(_lib.load().hx_mixctx_apply(self._h, _lib.ptr(cl), cl.shape[0], kind, _lib.ptr(out)))
        return out

    def set_bins(self, plan):
        """Bin the OUTPUT multipole of every matrix that ``binned`` builds from now on: ``plan`` is the ``binning.BinPlan`` of the axis
        (``BinPlan(np.arange(l1max + 1), edges, weights)`` for what ``mixing_matrices(bins=edges, weights=weights)`` asks for).  The
        binned Wigner-d tables are formed once per plan (hx_mixctx_set_bins)."""
        n1 = self.l1max + 1
        if plan.which.size != n1:
            raise ValueError(f"bin plan for {plan.which.size} multipoles, the matrices have {n1} rows")
        if plan.nbins < 1:
            raise ValueError("no bins")
        which = np.ascontiguousarray(plan.which, dtype=np.int32)
        w = np.ascontiguousarray(plan.w, dtype=np.float64)
        norm = np.ascontiguousarray(plan.norm, dtype=np.float64)
        _lib.check(_lib.load().hx_mixctx_set_bins(self._h, int(plan.nbins), which.ctypes.data, _lib.ptr(w), _lib.ptr(norm)))
        self.plan = plan

    def binned(self, cl, spin, out=None):
        """The matrices of ``self(cl, spin)`` with their rows binned by the plan of ``set_bins``: (nbins, l2max + 1), or
        (3, nbins, l2max + 1) for spin (2, 2) -- ``heracles.result.binned(Result(M, axis=-2), edges, weights).array`` without the
        full matrix ever being formed (hx_mixctx_apply_binned)."""
        if getattr(self, "plan", None) is None:
            raise ValueError("no bins set: call set_bins first")
        kind = self._kind(spin)
        cl = np.ascontiguousarray(np.asarray(cl, dtype=np.float64))
        shape = (self.plan.nbins, self.l2max + 1)
        out = _lib.result_array((3,) + shape if kind == 4 else shape, out)
        _lib.check(_lib.load().hx_mixctx_apply_binned(self._h, _lib.ptr(cl), cl.shape[0], kind, _lib.ptr(out)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().hx_mixctx_destroy(self._h)
            self._h = None
        self._buffers = {}  # (pinned arrays free themselves when their last view goes)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def _single_axis(value, what):
    """bins / weights of a result with ONE angular axis: the reference takes one entry or a 1-tuple (heracles/result.py:150-163)."""
    if isinstance(value, tuple):
        if len(value) != 1:
            raise ValueError(f"result and {what} have different number of ell axes")
        return value[0]
    return value


def request_cost(spin):
    """Relative cost of one mixing-matrix request: the number of (l, l') products behind it (an E/B key = 2, every other = 1)."""
    return 2 if all(spin) else 1


def split_requests(todo, rank, world):
    """The share of rank ``rank`` of ``world`` in a request list (``mixing_requests``): requests are independent (one product or two
    per key, no exchange: SURVEY section 8e, last bullet), so they are dealt by cost -- heaviest first, each to the rank with the least
    work so far, ties to the lowest rank -- and every rank keeps its share in the order of the list.  Deterministic: all ranks compute
    the same deal without talking to each other."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside a world of {world}")
    load = [0] * world
    owner = [0] * len(todo)
    for n in sorted(range(len(todo)), key=lambda n: (-request_cost(todo[n][2]), n)):
        r = min(range(world), key=lambda q: (load[q], q))
        owner[n] = r
        load[r] += request_cost(todo[n][2])
    return [req for n, req in enumerate(todo) if owner[n] == rank]


def mixing_matrices(fields, cls, *, l1max=None, l2max=None, l3max=None, bins=None, weights=None,
                    out=None, progress=None, context=None, rank=None, world=None):
    """Mixing matrices for fields from mask spectra (heracles/twopoint.py:316-401): same keys, order, spin dispatch and
    Result wrapping; the arithmetic goes through ONE MixmatContext per distinct (l1max, l2max, l3max), so the tables
    are built once for the whole job instead of once per matrix.

    ``bins`` / ``weights`` (heracles/twopoint.py:391-397, what heracles/cli.py:696-716 passes whenever the configuration has bins): the
    rows of every matrix binned as ``heracles.result.binned`` does along axis -2.  The binned rows are built DIRECTLY on the GPU from
    binned Wigner-d tables (``MixmatContext.set_bins`` / ``binned``): the full matrix is never formed and nbins x (l2max + 1) numbers per
    matrix leave the device.
    ``context`` (a callable (cl, l1max, l2max, l3max, spin) -> array) replaces the GPU path, e.g. to record the requests; its full
    matrices are binned on the host.
    ``rank`` / ``world``: this process computes only its share of the request list (``split_requests``: no collective; the union over the
    ranks is the single-process result)."""
    out = TocDict() if out is None else out
    progress = _NoProgress() if progress is None else progress
    todo = mixing_requests(fields, cls)
    if world is not None and world > 1:
        todo = split_requests(todo, 0 if rank is None else rank, world)
    pending = {}
    for target, ck, spin in todo:
        pending.setdefault(ck, []).append((target, spin))
    if bins is not None:
        bins, weights = _single_axis(bins, "bins"), _single_axis(weights, "weight")
    contexts, plans = {}, {}

    def plan_for(nrows):
        if nrows not in plans:
            plans[nrows] = BinPlan(np.arange(nrows), bins, weights)
        return plans[nrows]

    def compute(cl, spin):
        """the (binned) matrices of one request and the plan that binned them"""
        _, a, b, c = _mm_args(cl, l1max, l2max, l3max)
        if context is not None:
            mm = context(cl, a, b, c, spin)
            if bins is None:
                return mm, None
            plan = plan_for(np.shape(mm)[-2])
            return plan.apply(mm, np.ndim(mm) - 2), plan
        if (a, b, c) not in contexts:
            contexts[a, b, c] = MixmatContext(a, b, c)
            if bins is not None and plan_for(a + 1).nbins > 0:
                contexts[a, b, c].set_bins(plan_for(a + 1))
        ctx = contexts[a, b, c]
        if bins is None:
            return ctx(cl, spin), None
        if plan_for(a + 1).nbins == 0:  # (a single edge: no bin at all -- empty rows, as the reference returns them)
            plan = plan_for(a + 1)
            shape = (3, 0, b + 1) if MixmatContext._kind(spin) == 4 else (0, b + 1)
            return np.zeros(shape), plan
        return ctx.binned(cl, spin), ctx.plan

    try:
        for n, ck in enumerate(cls, start=1):
            progress.update(n, len(cls))
            for target, spin in pending.get(ck, ()):
                with progress.task(f"({target[0]}, {target[1]}, {target[2]}, {target[3]})"):
                    mm, plan = compute(np.asarray(cls[ck]), spin)
                    # second to last axis is the OUTPUT multipole (heracles/twopoint.py:391-393)
                    if plan is None:
                        mm = Result(mm, spin=spin, ell=np.arange(mm.shape[-2]), axis=-2)
                    else:
                        mm = wrap_binned(mm, spin, (mm.ndim - 2,), [plan])
                    out[target] = mm
    finally:
        for ctx in contexts.values():
            ctx.close()
    return out


# ---- heracles.twopoint.apply_mixing_matrix (heracles/twopoint.py:497-524) ---------------------------------------------
def _result_axis_array(result, name):
    """``get_result_array(result, name)[0]`` of heracles/result.py:53-72: the named angular array of the first ell axis, with the
    reference's defaults (ell = arange, lower = ell, upper = lower shifted by one, weight = ones)."""
    arr = getattr(result, name, None)
    n = result.shape[result.axis[0]]
    if arr is None:
        if name in ("ell", "lower"):
            src = getattr(result, "ell", None) if name == "lower" else None
            arr = np.arange(n) if src is None else src
        elif name == "upper":
            lo = _result_axis_array(result, "lower")
            arr = np.append(lo[1:], lo[-1] + 1)
        elif name == "weight":
            arr = np.ones(n)
        else:
            raise ValueError(f"cannot make default for array {name!r}")
    return arr[0] if isinstance(arr, tuple) else arr


def _matvec(M, xs):
    """rows of ``M @ x`` for every spectrum x of ``xs`` (hx_matvec: the matrix -- numpy array or device tensor / DeviceArray -- is
    read once for up to four spectra)."""
    M = getattr(M, "tensor", M)
    xs = np.ascontiguousarray(np.atleast_2d(xs), dtype=np.float64)
    n, m = M.shape
    if xs.shape[-1] != m:
        raise ValueError(f"matrix of shape {(n, m)} applied to spectra of length {xs.shape[-1]}")
    M = M.contiguous() if hasattr(M, "data_ptr") else np.ascontiguousarray(M, dtype=np.float64)
    out = np.empty((xs.shape[0], n))
    _lib.ensure_init()
    _lib.check(_lib.load().hx_matvec(n, m, _lib.ptr(M), xs.shape[0], _lib.ptr(xs), _lib.ptr(out)))
    return out


def apply_mixing_matrix(d, M):
    """Apply (inverse) mixing matrices to data spectra, key by key: ``heracles.twopoint.apply_mixing_matrix``
    (heracles/twopoint.py:497-524).  Spin-2 x spin-2 blocks use the three matrices of ``mixmat_eb`` as the reference does --
    EE' = M0 EE + M1 BB, BB' = M1 EE + M0 BB, EB' = M2 EB, BE' = M2 BE --, every other block is ``M @ cl`` per component
    spectrum; the angular arrays of the result follow the matrix' output axis.  The products run on the GPU (hx_matvec); matrices
    may be numpy arrays or device-resident (``DeviceArray`` / torch tensors)."""
    from dataclasses import replace

    out = {}
    for key, res in d.items():
        dtype = np.asarray(res.array).dtype
        s1, s2 = res.spin
        cl = np.atleast_2d(np.asarray(res.array))
        mm = M[key]
        mat = getattr(mm.array, "tensor", mm.array)  # (a DeviceArray's tensor: indexable per matrix)
        if s1 != 0 and s2 != 0:
            a = _matvec(mat[0], np.stack([cl[0, 0], cl[1, 1]]))   # M0 EE, M0 BB
            b = _matvec(mat[1], np.stack([cl[0, 0], cl[1, 1]]))   # M1 EE, M1 BB
            c = _matvec(mat[2], np.stack([cl[0, 1], cl[1, 0]]))   # M2 EB, M2 BE
            new = np.array([[a[0] + b[1], c[0]], [c[1], b[0] + a[1]]])
        else:
            new = np.squeeze(_matvec(mat, cl.reshape(-1, cl.shape[-1])))
        new = np.array(list(new), dtype=dtype)
        out[key] = replace(res, array=new, ell=_result_axis_array(mm, "ell"), lower=_result_axis_array(mm, "lower"),
                           upper=_result_axis_array(mm, "upper"), weight=_result_axis_array(mm, "weight"))
    return out


# ---- heracles.twopoint.invert_mixing_matrix (heracles/twopoint.py:404-494) ---------------------------------------------
def pinv(M, rcond=1e-5, *, device=None, info=False):
    """``np.linalg.pinv(M, rcond=rcond)`` on the GPU (``hx_pinv``: blocked one-sided Jacobi SVD, singular values <= rcond * the largest
    dropped).  ``M``: numpy array or device tensor (n, m); returns a numpy array (m, n), or a device tensor with ``device=``."""
    import ctypes as C_

    M = getattr(M, "tensor", M)
    n, m = M.shape
    M = M.contiguous() if hasattr(M, "data_ptr") else np.ascontiguousarray(M, dtype=np.float64)
    if device is None:
        out = np.empty((m, n))
    else:
        import torch

        out = torch.empty((m, n), dtype=torch.float64, device=device)
    inf = (C_.c_double * 4)()
    _lib.ensure_init()
    _lib.check(_lib.load().hx_pinv(int(n), int(m), _lib.ptr(M), float(rcond), _lib.ptr(out), inf))
    return (out, {"sweeps": int(inf[0]), "kept": int(inf[1]), "largest": inf[2], "smallest_kept": inf[3]}) if info else out


def invert_mixing_matrix(M, rcond=1e-5, progress=None):
    """Pseudo-inverses of mixing matrices, key by key: ``heracles.twopoint.invert_mixing_matrix`` (heracles/twopoint.py:404-494).  Spin-2 x
    spin-2 keys hold the three matrices of ``mixmat_eb``: M0 +- M1 are inverted separately (the transformation to Cl^EE +- Cl^BB makes
    the E/B system block diagonal), [0] = (P + Q) / 2, [1] = (P - Q) / 2 of their inverses P, Q, [2] = pinv(M2); every other key is
    ``pinv(M)``.  ``rcond``: a number or a mapping key -> number (a key missing from it is a ``KeyError``, as in the reference).  A
    non-square matrix swaps its ell axes: the angular arrays of the result are then those of the inverse's output axis (0 .. size).
    Every ``pinv`` runs on the GPU (``hx_pinv``)."""
    from collections.abc import Mapping
    from dataclasses import replace

    prog = progress if progress is not None else _NoProgress()
    out = {}
    for count, (key, value) in enumerate(M.items(), 1):
        prog.update(count, len(M))
        mat = getattr(value.array, "tensor", value.array)
        s1, s2 = value.spin
        n, m = mat.shape[-2], mat.shape[-1]
        if isinstance(rcond, Mapping):
            if key not in rcond:
                raise KeyError(f"Missing rcond value for wm key: {key}")
            rc = rcond[key]
        else:
            rc = rcond
        with prog.task(f"invert {key}"):
            if s1 != 0 and s2 != 0:
                plus, minus = pinv(mat[0] + mat[1], rc), pinv(mat[0] - mat[1], rc)
                inv = np.array([(plus + minus) / 2, (plus - minus) / 2, pinv(mat[2], rc)])
            else:
                inv = pinv(mat, rc)
        if n != m:
            size = inv.shape[value.axis[0]]
            out[key] = replace(value, array=inv, ell=np.arange(size), lower=np.arange(size), upper=np.arange(1, size + 1), weight=np.ones(size))
        else:
            out[key] = replace(value, array=inv)
    return out
