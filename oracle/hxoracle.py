"""ctypes front-end of the CPU oracle (oracle/hx_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under heracles_amd/ may import this module.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = C.POINTER(C.c_double)


def build(force: bool = False) -> str:
    """Compile liboracle if needed (gcc only; no GPU involved).  HXORACLE_LIB selects another build of the same source
    (the AddressSanitizer / UBSan build of `make -C oracle asan`, tests/test_oracle_sanitizers.py)."""
    if os.environ.get("HXORACLE_LIB"):
        return os.environ["HXORACLE_LIB"]
    so = os.path.join(_HERE, "libhxoracle.so")
    src = os.path.join(_HERE, "hx_oracle.c")
    hdr = os.path.join(_HERE, "hx_oracle.h")
    if (
        force
        or not os.path.exists(so)
        or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.hxo_nlm.restype = C.c_int64
        L.hxo_map2alm.restype = C.c_int
        L.hxo_alm2map.restype = C.c_int
        L.hxo_wigner3j_l3.restype = C.c_int
        L.hxo_num_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def nlm(lmax: int) -> int:
    return (lmax + 1) * (lmax + 2) // 2


def num_threads() -> int:
    return lib().hxo_num_threads()


def set_mstride(s: int):
    """bench sampling knob: Legendre stage of map2alm processes only every s-th m."""
    lib().hxo_set_mstride(C.c_int(int(s)))


def last_timings():
    a, b = C.c_double(), C.c_double()
    lib().hxo_last_timings(C.byref(a), C.byref(b))
    return a.value, b.value


def ring_info(nside: int, ring: int):
    sp = C.c_int64()
    nphi = C.c_int()
    z = C.c_double()
    sth = C.c_double()
    phi0 = C.c_double()
    lib().hxo_ring_info(
        C.c_int(nside), C.c_int(ring), C.byref(sp), C.byref(nphi), C.byref(z),
        C.byref(sth), C.byref(phi0),
    )
    return sp.value, nphi.value, z.value, sth.value, phi0.value


def pix2ang(nside: int):
    """(theta, phi) of all RING pixel centres, from the oracle's ring table."""
    npix = 12 * nside * nside
    theta = np.empty(npix)
    phi = np.empty(npix)
    for ring in range(1, 4 * nside):
        sp, nphi, z, sth, phi0 = ring_info(nside, ring)
        theta[sp : sp + nphi] = np.arctan2(sth, z)
        phi[sp : sp + nphi] = phi0 + 2 * np.pi * np.arange(nphi) / nphi
    return theta, phi


def map2alm(maps, nside, lmax, spin=0, ring_weights=None, pix_weights=None, niter=0,
            use_fft=True):
    maps = np.ascontiguousarray(maps, dtype=np.float64)
    npix = 12 * nside * nside
    lead = maps.shape[:-1]
    m2 = maps.reshape(-1, npix)
    ncomp = m2.shape[0]
    alms = np.zeros((ncomp, nlm(lmax)), dtype=np.complex128)
    rw = None if ring_weights is None else np.ascontiguousarray(ring_weights, dtype=np.float64)
    pw = None if pix_weights is None else np.ascontiguousarray(pix_weights, dtype=np.float64)
    rc = lib().hxo_map2alm(
        C.c_int(nside), C.c_int(lmax), C.c_int(spin), C.c_int(ncomp), _p(m2), _p(alms),
        _p(rw) if rw is not None else None, _p(pw) if pw is not None else None,
        C.c_int(niter), C.c_int(1 if use_fft else 0),
    )
    if rc != 0:
        raise ValueError(f"hxo_map2alm failed ({rc})")
    return alms.reshape(*lead, -1)


def points2alm(theta, phi, values, lmax, spin=0):
    """Direct sum alm = sum_p values_p conj(sY_lm(theta_p, phi_p)) (heracles/ducc.py:121-128): values (ncomp, npoints)."""
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    phi = np.ascontiguousarray(phi, dtype=np.float64)
    v2 = np.ascontiguousarray(values, dtype=np.float64).reshape(-1, theta.size)
    alms = np.zeros((v2.shape[0], nlm(lmax)), dtype=np.complex128)
    rc = lib().hxo_points2alm(C.c_int(lmax), C.c_int(spin), C.c_int(v2.shape[0]), C.c_int64(theta.size), _p(theta), _p(phi),
                              _p(v2), _p(alms))
    if rc != 0:
        raise ValueError(f"hxo_points2alm failed ({rc})")
    return alms


def alm2map(alms, nside, lmax, spin=0, use_fft=True):
    alms = np.ascontiguousarray(alms, dtype=np.complex128)
    lead = alms.shape[:-1]
    a2 = alms.reshape(-1, nlm(lmax))
    ncomp = a2.shape[0]
    npix = 12 * nside * nside
    maps = np.zeros((ncomp, npix))
    rc = lib().hxo_alm2map(
        C.c_int(nside), C.c_int(lmax), C.c_int(spin), C.c_int(ncomp), _p(a2), _p(maps),
        C.c_int(1 if use_fft else 0),
    )
    if rc != 0:
        raise ValueError(f"hxo_alm2map failed ({rc})")
    return maps.reshape(*lead, npix)


def alm2lmax(n: int) -> int:
    """heracles/twopoint.py:55-60"""
    return (int((8 * n + 1) ** 0.5 + 0.01) - 3) // 2


def alm2cl(alm, alm2=None, *, lmax=None):
    """Block alm2cl with the reference's broadcasting (twopoint.py:63-101)."""
    if alm2 is None:
        alm2 = alm
    alm = np.ascontiguousarray(alm, dtype=np.complex128)
    alm2 = np.ascontiguousarray(alm2, dtype=np.complex128)
    l1, l2 = alm2lmax(alm.shape[-1]), alm2lmax(alm2.shape[-1])
    lout = min(l1, l2) if lmax is None else min(lmax, l1, l2)
    a = alm.reshape(-1, alm.shape[-1])
    b = alm2.reshape(-1, alm2.shape[-1])
    out = np.empty((a.shape[0], b.shape[0], lout + 1))
    tmp = np.empty(lout + 1)
    for i in range(a.shape[0]):
        for j in range(b.shape[0]):
            lib().hxo_alm2cl(_p(a[i]), C.c_int(l1), _p(b[j]), C.c_int(l2), C.c_int(lout), _p(tmp))
            out[i, j] = tmp
    return out.reshape(*alm.shape[:-1], *alm2.shape[:-1], lout + 1)


def gauss_legendre(n: int):
    x = np.empty(n)
    w = np.empty(n)
    lib().hxo_gauss_legendre(C.c_int(n), _p(x), _p(w))
    return x, w


def wigner_d(lmax: int, a: int, b: int, x: float):
    out = np.empty(lmax + 1)
    lib().hxo_wigner_d(C.c_int(lmax), C.c_int(a), C.c_int(b), C.c_double(x), _p(out))
    return out


def legendre_funcs(lmax: int, x: float):
    P = np.empty(lmax + 1)
    dP = np.empty(lmax + 1)
    d20 = np.empty(max(lmax - 1, 0))
    d22 = np.empty(max(lmax - 1, 0))
    d2m2 = np.empty(max(lmax - 1, 0))
    lib().hxo_legendre_funcs(C.c_int(lmax), C.c_double(x), _p(P), _p(dP), _p(d20), _p(d22), _p(d2m2))
    return (P, dP), (d20, d22, d2m2)


def cl2corr(cls, lmax=None):
    cls = np.asarray(cls, dtype=np.float64)
    if cls.ndim == 1:
        cls = np.stack([cls, np.zeros_like(cls), np.zeros_like(cls), np.zeros_like(cls)]).T
    if lmax is None:
        lmax = cls.shape[0] - 1
    cls = np.ascontiguousarray(cls[: lmax + 1])
    x, w = gauss_legendre(lmax + 1)
    xw = np.concatenate([x, w])
    out = np.empty((lmax + 1, 4))
    lib().hxo_cl2corr(C.c_int(lmax), _p(cls), _p(xw), _p(out))
    return out


def corr2cl(corrs, lmax=None):
    corrs = np.asarray(corrs, dtype=np.float64)
    if corrs.ndim == 1:
        corrs = np.stack([corrs, np.zeros_like(corrs), np.zeros_like(corrs), np.zeros_like(corrs)]).T
    if lmax is None:
        lmax = corrs.shape[0] - 1
    corrs = np.ascontiguousarray(corrs)
    x, w = gauss_legendre(lmax + 1)
    xw = np.concatenate([x, w])
    out = np.empty((lmax + 1, 4))
    lib().hxo_corr2cl(C.c_int(lmax), _p(corrs), _p(xw), _p(out))
    return out


def wigner3j_l3(l1, l2, m1, m2):
    out = np.zeros(l1 + l2 + 2)
    n = C.c_int()
    jmin = lib().hxo_wigner3j_l3(C.c_int(l1), C.c_int(l2), C.c_int(m1), C.c_int(m2), _p(out), C.byref(n))
    return jmin, out[: n.value].copy()


def _mm_defaults(cl, l1max, l2max, l3max):
    # defaults of the third-party convolvecl.mixmat are not in the reference tree
    # [UNVERIFIED]; chosen so that tests/test_twopoint.py:412-462 shapes hold.
    cl = np.ascontiguousarray(cl, dtype=np.float64)
    if l3max is None:
        l3max = cl.shape[-1] - 1
    if l1max is None:
        l1max = l3max
    if l2max is None:
        l2max = l1max
    if cl.shape[-1] < l3max + 1:
        cl = np.concatenate([cl, np.zeros(l3max + 1 - cl.shape[-1])])
    return cl, l1max, l2max, l3max


def mixmat(cl, l1max=None, l2max=None, l3max=None, spin=(0, 0)):
    cl, l1max, l2max, l3max = _mm_defaults(cl, l1max, l2max, l3max)
    out = np.empty((l1max + 1, l2max + 1))
    lib().hxo_mixmat(_p(cl), C.c_int(l1max), C.c_int(l2max), C.c_int(l3max),
                     C.c_int(spin[0]), C.c_int(spin[1]), _p(out))
    return out


def mixmat_eb(cl, l1max=None, l2max=None, l3max=None, spin=(2, 2)):
    cl, l1max, l2max, l3max = _mm_defaults(cl, l1max, l2max, l3max)
    out = np.empty((3, l1max + 1, l2max + 1))
    lib().hxo_mixmat_eb(_p(cl), C.c_int(l1max), C.c_int(l2max), C.c_int(l3max), _p(out))
    return out


def mixmat_block(cl, rows, cols, l3max=None, spin=(0, 0)):
    """Block rows = (l1lo, l1hi), cols = (l2lo, l2hi) (inclusive) of ``mixmat(cl, l3max=l3max, spin=spin)``."""
    cl = np.ascontiguousarray(cl, dtype=np.float64)
    if l3max is None:
        l3max = cl.shape[-1] - 1
    if cl.shape[-1] < l3max + 1:
        cl = np.concatenate([cl, np.zeros(l3max + 1 - cl.shape[-1])])
    out = np.empty((rows[1] - rows[0] + 1, cols[1] - cols[0] + 1))
    lib().hxo_mixmat_block(_p(cl), C.c_int(rows[0]), C.c_int(rows[1]), C.c_int(cols[0]), C.c_int(cols[1]),
                           C.c_int(l3max), C.c_int(spin[0]), C.c_int(spin[1]), _p(out))
    return out


def mixmat_eb_block(cl, rows, cols, l3max=None):
    """The same block of the three matrices of ``mixmat_eb``: shape (3, rows, cols)."""
    cl = np.ascontiguousarray(cl, dtype=np.float64)
    if l3max is None:
        l3max = cl.shape[-1] - 1
    if cl.shape[-1] < l3max + 1:
        cl = np.concatenate([cl, np.zeros(l3max + 1 - cl.shape[-1])])
    out = np.empty((3, rows[1] - rows[0] + 1, cols[1] - cols[0] + 1))
    lib().hxo_mixmat_eb_block(_p(cl), C.c_int(rows[0]), C.c_int(rows[1]), C.c_int(cols[0]), C.c_int(cols[1]),
                              C.c_int(l3max), _p(out))
    return out


# ---- catalogue -> map accumulation (numpy restatement) -------------------------------
def ang2pix_ring(nside, lon, lat):
    """hp.ang2pix(nside, lon, lat, lonlat=True) as called at heracles/healpy.py:157.
    healpy (1.16) is not vendored in /root/reference; this restates its published code
    path: healpy lonlat2thetaphi (theta = pi/2 - radians(lat), phi = radians(lon)), then
    healpix_cxx T_Healpix_Base::ang2pix -> loc2pix, RING branch (Gorski et al. 2005).
    Parity with healpy itself is unpinned; pinned here on the pix2ang round trip."""
    lon = np.asarray(lon, dtype=np.float64)
    lat = np.asarray(lat, dtype=np.float64)
    nside = int(nside)
    theta = np.pi / 2.0 - np.radians(lat)
    phi = np.radians(lon)
    z = np.cos(theta)
    have_sth = (theta < 0.01) | (theta > 3.14159 - 0.01)
    sth = np.where(have_sth, np.sin(theta), 0.0)
    za = np.abs(z)
    v = phi * 0.6366197723675813430755350534900574
    with np.errstate(invalid="ignore"):
        neg = np.fmod(v, 4.0) + 4.0
        tt = np.where(v >= 0, np.where(v < 4.0, v, np.fmod(v, 4.0)), np.where(neg == 4.0, 0.0, neg))
    nl4 = 4 * nside
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    # equatorial belt
    temp1 = nside * (0.5 + tt)
    temp2 = nside * z * 0.75
    jp = (temp1 - temp2).astype(np.int64)
    jm = (temp1 + temp2).astype(np.int64)
    ir = nside + 1 + jp - jm
    kshift = 1 - (ir & 1)
    t1 = jp + jm - nside + kshift + 1 + nl4 + nl4
    ip = (t1 >> 1) % nl4
    pe = ncap + (ir - 1) * nl4 + ip
    # polar caps
    tp = tt - tt.astype(np.int64)
    with np.errstate(invalid="ignore", divide="ignore"):
        tmp = np.where((za < 0.99) | ~have_sth, nside * np.sqrt(3.0 * (1.0 - za)), nside * sth / np.sqrt((1.0 + za) / 3.0))
    jp2 = (tp * tmp).astype(np.int64)
    jm2 = ((1.0 - tp) * tmp).astype(np.int64)
    ir2 = jp2 + jm2 + 1
    ip2 = np.minimum((tt * ir2).astype(np.int64), 4 * ir2 - 1)
    pn = 2 * ir2 * (ir2 - 1) + ip2
    ps = npix - 2 * ir2 * (ir2 + 1) + ip2
    return np.where(za <= 2.0 / 3.0, pe, np.where(z > 0, pn, ps))


def map_values(nside, lon, lat, data, values):
    """HealpixMapper.map_values (heracles/healpy.py:144-160): the sequential loop of
    heracles/healpy.py:58-66, maps[..., i] += values[..., j] in catalogue order.
    np.add.at is that same unbuffered in-order accumulation (the reference's own test
    uses it as the expectation, tests/test_healpy.py:57-77)."""
    ipix = ang2pix_ring(nside, lon, lat)
    flat = data.reshape(-1, data.shape[-1])
    vals = np.broadcast_to(np.asarray(values, dtype=np.float64), (*data.shape[:-1], ipix.size)).reshape(-1, ipix.size)
    for row in range(flat.shape[0]):
        np.add.at(flat[row], ipix, vals[row])


# ---- resolution change (numpy restatement of healpy.ud_grade) ------------------------
_JRLL = np.array([2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4], dtype=np.int64)
_JPLL = np.array([1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7], dtype=np.int64)
UNSEEN = -1.6375e30


def _spread(v):
    v = np.asarray(v, dtype=np.int64)
    out = np.zeros_like(v)
    for b in range(16):
        out |= ((v >> b) & 1) << (2 * b)
    return out


def _compress(v):
    v = np.asarray(v, dtype=np.int64)
    out = np.zeros_like(v)
    for b in range(16):
        out |= ((v >> (2 * b)) & 1) << b
    return out


def _isqrt(v):
    return np.floor(np.sqrt(np.asarray(v, dtype=np.float64) + 0.5)).astype(np.int64)


def nest2ring(nside, pnest):
    """healpix_cxx T_Healpix_Base::nest2ring (nest2xyf + xyf2ring); nside a power of two."""
    pnest = np.asarray(pnest, dtype=np.int64)
    order = int(nside).bit_length() - 1
    npface, npix, ncap, nl4 = nside * nside, 12 * nside * nside, 2 * nside * (nside - 1), 4 * nside
    face = pnest >> (2 * order)
    pf = pnest & (npface - 1)
    ix, iy = _compress(pf), _compress(pf >> 1)
    jr = (_JRLL[face] << order) - ix - iy - 1
    north, south = jr < nside, jr > 3 * nside
    nr = np.where(north, jr, np.where(south, nl4 - jr, nside))
    n_before = np.where(north, 2 * nr * (nr - 1), np.where(south, npix - 2 * (nr + 1) * nr, ncap + (jr - nside) * nl4))
    kshift = np.where(north | south, 0, (jr - nside) & 1)
    jp = (_JPLL[face] * nr + ix - iy + 1 + kshift) // 2
    jp = np.where(jp > nl4, jp - nl4, np.where(jp < 1, jp + nl4, jp))
    return n_before + jp - 1


def ring2nest(nside, pring):
    """healpix_cxx T_Healpix_Base::ring2nest (ring2xyf + xyf2nest); nside a power of two."""
    pring = np.asarray(pring, dtype=np.int64)
    order = int(nside).bit_length() - 1
    npface, npix, ncap, nl4 = nside * nside, 12 * nside * nside, 2 * nside * (nside - 1), 4 * nside
    # north cap
    irn = (1 + _isqrt(1 + 2 * pring)) >> 1
    ipn = (pring + 1) - 2 * irn * (irn - 1)
    fn = (ipn - 1) // np.maximum(irn, 1)
    # equatorial belt
    ip = pring - ncap
    tmp = ip >> (order + 2)
    ire_ring = tmp + nside
    ipe = ip - tmp * nl4 + 1
    kse = (ire_ring + nside) & 1
    ire, irm = tmp + 1, 2 * nside + 1 - tmp
    ifm = (ipe - ire // 2 + nside - 1) >> order
    ifp = (ipe - irm // 2 + nside - 1) >> order
    fe = np.where(ifp == ifm, ifp | 4, np.where(ifp < ifm, ifp, ifm + 8))
    # south cap
    ips = npix - pring
    irs = (1 + _isqrt(np.maximum(2 * ips - 1, 0))) >> 1
    iphs = 4 * irs + 1 - (ips - 2 * irs * (irs - 1))
    fs = 8 + (iphs - 1) // np.maximum(irs, 1)
    north, south = pring < ncap, pring >= npix - ncap
    iring = np.where(north, irn, np.where(south, nl4 - irs, ire_ring))
    iphi = np.where(north, ipn, np.where(south, iphs, ipe))
    kshift = np.where(north | south, 0, kse)
    nr = np.where(north, irn, np.where(south, irs, nside))
    face = np.where(north, fn, np.where(south, fs, fe))
    irt = iring - _JRLL[face] * nside + 1
    ipt = 2 * iphi - _JPLL[face] * nr - kshift - 1
    ipt = np.where(ipt >= 2 * nside, ipt - 8 * nside, ipt)
    ix, iy = (ipt - irt) >> 1, (-ipt - irt) >> 1
    return face * npface + _spread(ix) + (_spread(iy) << 1)


def ud_grade(m, nside_out):
    """hp.ud_grade(m, nside_out, dtype=float64) as called at heracles/healpy.py:205-209
    (RING in, RING out, pess=False, power=None): healpy.pixelfunc.ud_grade = reorder to
    NEST, _ud_grade_core, reorder back.  np.sum(axis=1) below IS the arithmetic healpy
    runs (pairwise summation of each parent's children in NEST order)."""
    m = np.asarray(m, dtype=np.float64)
    if m.ndim > 1:
        return np.stack([ud_grade(x, nside_out) for x in m])
    npix_in = m.shape[-1]
    nside_in = int(round(np.sqrt(npix_in / 12)))
    for ns in (nside_in, nside_out):
        if ns < 1 or ns & (ns - 1):
            raise ValueError(f"{ns} is not a valid nside parameter (must be a power of 2)")
    npix_out = 12 * nside_out * nside_out
    m_nest = m[nest2ring(nside_in, np.arange(npix_in))]
    if nside_out > nside_in:
        rat2 = npix_out // npix_in
        out_nest = np.outer(m_nest, np.ones(rat2)).reshape(npix_out)
    elif nside_out < nside_in:
        rat2 = npix_in // npix_out
        mr = m_nest.reshape(npix_out, rat2)
        with np.errstate(invalid="ignore"):
            goods = ~((np.absolute(mr - UNSEEN) <= 1e-15 + 1e-5 * np.absolute(UNSEEN)) | (~np.isfinite(mr)))
            out_nest = np.sum(mr * goods, axis=1)
        nhit = goods.sum(axis=1)
        out_nest[nhit != 0] = out_nest[nhit != 0] / nhit[nhit != 0]
        out_nest[nhit == 0] = UNSEEN
    else:
        out_nest = m_nest
    return out_nest[ring2nest(nside_out, np.arange(npix_out))]


def expand_full_weights(nside, wgt):
    """Full-sky multiplicative pixel weights ``1 + w`` from healpy's compressed half-quadrant weights (the values in
    ``healpix_full_weights_nside_NNNN.fits``), RING order: restatement of the published expansion of healpix_cxx's
    ``apply_fullweights`` (third-party; not in /root/reference -- "parity unpinned" against healpy itself), ring by ring with
    plain Python loops."""
    wgt = np.asarray(wgt, dtype=np.float64)
    assert wgt.size == ((nside + 1) * (3 * nside + 1)) // 4
    npix = 12 * nside * nside
    out = np.zeros(npix)
    pix = vpix = 0
    for i in range(2 * nside):
        shifted = (i < nside - 1) or bool((i + nside) & 1)
        qpix = min(nside, i + 1)
        odd = qpix & 1
        wpix = ((qpix + 1) >> 1) + (0 if (odd or shifted) else 1)
        psouth = npix - pix - (qpix << 2)
        for j in range(qpix << 2):
            j4 = j % qpix
            rpix = min(j4, qpix - (1 if shifted else 0) - j4)
            out[pix + j] = 1.0 + wgt[vpix + rpix]
            if i != 2 * nside - 1:
                out[psouth + j] = 1.0 + wgt[vpix + rpix]
        pix += qpix << 2
        vpix += wpix
    assert vpix == wgt.size and (out != 0).all()
    return out


def fourier_analysis(maps, nside, lmax, pix_weights=None):
    """Ring Fourier stage of map2alm alone: F[comp][ring][m], all 4 nside - 1 rings, m <= lmax (checker of the m-sharded route)."""
    maps = np.ascontiguousarray(maps, dtype=np.float64).reshape(-1, 12 * nside * nside)
    F = np.zeros((maps.shape[0], 4 * nside - 1, lmax + 1), dtype=np.complex128)
    pw = None if pix_weights is None else np.ascontiguousarray(pix_weights, dtype=np.float64)
    lib().hxo_fourier_analysis(C.c_int(nside), C.c_int(lmax), C.c_int(maps.shape[0]), _p(maps), _p(pw) if pw is not None else None, _p(F))
    return F


def legendre_analysis(F, nside, lmax, spin=0, ring_weights=None):
    """Legendre stage of map2alm alone on a given F[comp][ring][m]: alm[comp][nlm]."""
    F = np.ascontiguousarray(F, dtype=np.complex128)
    alms = np.zeros((F.shape[0], nlm(lmax)), dtype=np.complex128)
    rw = None if ring_weights is None else np.ascontiguousarray(ring_weights, dtype=np.float64)
    rc = lib().hxo_legendre_analysis(C.c_int(nside), C.c_int(lmax), C.c_int(spin), C.c_int(F.shape[0]), _p(F), _p(rw) if rw is not None else None,
                                     _p(alms))
    if rc != 0:
        raise ValueError(f"hxo_legendre_analysis failed ({rc})")
    return alms
