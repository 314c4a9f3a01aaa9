"""ctypes binding of oracle/libhxcpufast.so: the VECTORISED CPU restatement of HEALPix map2alm (hx_cpu_fast.c) that bench.py's
``cpu_baseline`` leg times on the host cores.  Measurement infrastructure, like the oracle: nothing under heracles_amd/ imports it.
The scalar oracle (hxoracle.py) checks it (tests/test_oracle_fast.py)."""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libhxcpufast.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "hx_cpu_fast.c")):
            try:
                subprocess.check_call(["make", "-C", _HERE, "-s", "libhxcpufast.so"])
            except (OSError, subprocess.CalledProcessError):
                if not os.path.exists(_LIB):  # (a prebuilt library whose time stamp did not survive a copy is still the library)
                    raise
        L = C.CDLL(_LIB)
        L.hxf_map2alm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def supported() -> bool:
    """AVX-512 F + DQ on this CPU"""
    return bool(lib().hxf_supported())


def num_threads() -> int:
    return int(lib().hxf_num_threads())


def map2alm(maps, nside, lmax, spin=0, pix_weights=None):
    """maps (ncomp, npix) RING float64 -> (alms (ncomp, nlm) complex128 m-major, (ring_stage_seconds, legendre_seconds))."""
    maps = np.ascontiguousarray(np.atleast_2d(maps), dtype=np.float64)
    ncomp, npix = maps.shape
    if npix != 12 * nside * nside:
        raise ValueError("maps do not have 12 nside^2 pixels")
    pw = None if pix_weights is None else np.ascontiguousarray(pix_weights, dtype=np.float64)
    alms = np.empty((ncomp, (lmax + 1) * (lmax + 2) // 2), dtype=np.complex128)
    tim = (C.c_double * 2)()
    rc = lib().hxf_map2alm(int(nside), int(lmax), int(spin), ncomp, maps.ctypes.data, None if pw is None else pw.ctypes.data, alms.ctypes.data, tim)
    if rc:
        raise RuntimeError({-1: "bad arguments", -2: "out of memory", -3: "no AVX-512 on this CPU"}.get(rc, f"error {rc}"))
    return alms, (tim[0], tim[1])


def alm2cl(a, b):
    """Spectrum of two alm arrays of the same lmax (threaded over m)."""
    a = np.ascontiguousarray(a, dtype=np.complex128)
    b = np.ascontiguousarray(b, dtype=np.complex128)
    n = a.shape[-1]
    lmax = (int((8 * n + 1) ** 0.5 + 0.01) - 3) // 2
    cl = np.empty(lmax + 1)
    lib().hxf_alm2cl.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib().hxf_alm2cl.restype = None
    lib().hxf_alm2cl(a.ctypes.data, b.ctypes.data, lmax, cl.ctypes.data)
    return cl


def set_threads(n: int):
    """OpenMP threads from now on (shared with the oracle's library: one OpenMP runtime per process)."""
    lib().hxf_set_threads(int(n))


def cpu_quota():
    """CPUs this process may actually use: the cgroup CPU quota (cpu.max / cfs_quota_us) if one is set, capped by the affinity mask.
    A one-GPU lease of an 8-GPU host sees all 256 logical CPUs but is throttled to its share: more OpenMP threads than that only burn
    the quota in spin-waits."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = float(f.read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota
