"""Shared helpers for the parity tests (independent brute-force references)."""

from math import comb, factorial

import numpy as np


def idx(lmax, l, m):
    return m * (2 * lmax + 1 - m) // 2 + l


def random_alm(rng, lmax, lmin=0, shape=()):
    nlm = (lmax + 1) * (lmax + 2) // 2
    a = rng.standard_normal(shape + (nlm,)) + 1j * rng.standard_normal(shape + (nlm,))
    a[..., : lmax + 1] = a[..., : lmax + 1].real
    for m in range(lmax + 1):
        for l in range(m, min(lmin, lmax + 1)):
            a[..., idx(lmax, l, m)] = 0
    return a


def sYlm(s, l, m, th, ph):
    """Spin-weighted spherical harmonic, explicit finite sum (Goldberg et al. 1967):
    independent of every recursion used in the oracle and in the HIP kernels."""
    pref = (-1.0) ** (l + m - s) * np.sqrt(
        factorial(l + m) * factorial(l - m) * (2 * l + 1) / (4 * np.pi * factorial(l + s) * factorial(l - s))
    )
    out = np.zeros_like(th)
    for r in range(0, l - s + 1):
        if r + s - m < 0 or r + s - m > l + s:
            continue
        out += (-1.0) ** r * comb(l - s, r) * comb(l + s, r + s - m) * (np.cos(th / 2) / np.sin(th / 2)) ** (2 * r + s - m)
    return pref * np.sin(th / 2) ** (2 * l) * out * np.exp(1j * m * ph)


def key_str(key):
    return "|".join(str(k) for k in key)


def band_limited_maps(oracle, rng, nside, lmax, spin, nfield):
    """Synthesise band-limited maps with the oracle: returns (alms, maps)."""
    if spin == 0:
        alm = random_alm(rng, lmax, 0, (nfield,))
        cl = 1.0 / (1.0 + np.arange(lmax + 1)) ** 2
    else:
        alm = random_alm(rng, lmax, 2, (2 * nfield,))
        cl = 1.0 / (1.0 + np.arange(lmax + 1)) ** 2
    for m in range(lmax + 1):
        s = idx(lmax, m, m)
        alm[..., s : s + lmax - m + 1] *= np.sqrt(cl[m:])
    maps = oracle.alm2map(alm, nside, lmax, spin=spin)
    return alm, maps


def lambda_lm_column(m, lmax, x, sth):
    """lambda_lm(theta) = Y_lm(theta, 0) for l = m..lmax on an array of co-latitudes, by the textbook
    normalised three-term recursion in EXTENDED precision (np.longdouble: 64-bit mantissa and a
    2^+-16384 range: sin^m(theta) stays a normal number down to 1e-4932, and a term below that is zero
    at any precision that matters).
    Independent of the recursions in the oracle and the kernels (two-step / Wigner-d forms)."""
    ld = np.longdouble
    x, sth = np.asarray(x, dtype=ld), np.asarray(sth, dtype=ld)
    k = np.arange(1, m + 1, dtype=ld)
    lognorm = ld(0.5) * (np.log(ld(2 * m + 1)) - np.log(ld(4) * ld(np.pi)) + np.sum(np.log((2 * k - 1) / (2 * k))))
    with np.errstate(divide="ignore"):
        cur = (ld(-1.0) if m & 1 else ld(1.0)) * np.exp(lognorm + ld(m) * np.log(sth))
    out = np.empty((lmax - m + 1,) + x.shape, dtype=ld)
    out[0] = cur
    prev = np.zeros_like(cur)
    for l in range(m, lmax):
        l1 = ld(l + 1)
        a = np.sqrt((4 * l1 * l1 - 1) / (l1 * l1 - ld(m) * m))
        b = np.sqrt((ld(l) * l - ld(m) * m) / (4 * ld(l) * l - 1)) if l > 0 else ld(0)
        nxt = a * (x * cur - b * prev)
        prev, cur = cur, nxt
        out[l + 1 - m] = cur
    return out
