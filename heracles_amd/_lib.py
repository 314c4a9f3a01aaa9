"""ctypes binding of libhxsht.so (include/hxsht.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 device is
usable, every compute call raises :class:`HxError`.
"""

from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HX_LIBRARY selects another build of the same library (kernel timing experiments)
_LIBNAME = os.environ.get("HX_LIBRARY") or os.path.join(_HERE, "libhxsht.so")
_lock = threading.Lock()
_lib = None

HX_OK = 0
HX_ERR_ARG, HX_ERR_NO_DEVICE, HX_ERR_HIP, HX_ERR_MEM, HX_ERR_UNSUPPORTED = -1, -2, -3, -4, -5

#: every symbol declared in include/hxsht.h
SYMBOLS = (
    "hx_version", "hx_last_error", "hx_device_count", "hx_init", "hx_set_stream",
    "hx_get_stream", "hx_set_async", "hx_synchronize", "hx_timer_start", "hx_timer_stop",
    "hx_profile_enable", "hx_profile_reset", "hx_profile_get", "hx_plan_create", "hx_set_max_lds_fft",
    "hx_plan_destroy", "hx_plan_scratch_bytes", "hx_plan_release_scratch", "hx_set_scratch_budget", "hx_plan_last_chunks", "hx_plan_mfma_flops", "hx_plan_executed_flops", "hx_executed_flops", "hx_measure_peaks", "hx_measured_mfma_clock", "hx_map2alm", "hx_map2alm_multi", "hx_map2alm_list", "hx_alm2map", "hx_copy",
    "hx_alm2cl_pairs", "hx_alm2cl_pairs_range", "hx_gauss_legendre", "hx_gauss_legendre_dd", "hx_wigner_d_table", "hx_mixmat",
    "hx_mixmat_eb", "hx_mixmat_batch", "hx_mixctx_create", "hx_mixctx_apply", "hx_mixctx_destroy", "hx_mixctx_set_bins", "hx_mixctx_apply_binned", "hx_cl2corr", "hx_corr2cl", "hx_ang2pix_ring", "hx_map_values", "hx_ud_grade", "hx_reorder", "hx_matvec", "hx_pinv", "hx_alm_resample", "hx_region_maps", "hx_alm_subtract", "hx_fits_unpack_f64", "hx_fits_pack_f64",
    "hx_pointsht_create", "hx_pointsht_destroy", "hx_pointsht_info", "hx_pointsht_adjoint",
    "hx_pixel_weights_size", "hx_pixel_weights_expand",
    "hx_ring_modes_size", "hx_ring_modes", "hx_legendre_from_modes", "hx_allgather_alms", "hx_host_alloc", "hx_host_free", "hx_mixmat_gemm_clock", "hx_mixmat_release", "hx_release_caches",
)


class HxError(RuntimeError):
    """Error reported by libhxsht (code, message)."""

    def __init__(self, code, msg):
        super().__init__(f"libhxsht error {code}: {msg}")
        self.code = code


def library_path() -> str:
    return _LIBNAME


def load():
    """Load libhxsht.so (built in-tree by __graft_entry__.build() / make -C csrc)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIBNAME):
            raise HxError(
                HX_ERR_NO_DEVICE,
                f"{_LIBNAME} not found: build it with `make -C heracles_amd/csrc` "
                "(hipcc, gfx950).  heracles_amd has no CPU fallback.",
            )
        try:
            # one HIP runtime per process: torch bundles its own libamdhip64 under the same
            # soname, so it has to be the copy already resident when libhxsht.so is loaded
            # (loading the system runtime first leaves torch without a visible device)
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_LIBNAME)
        vp, i, dp = C.c_void_p, C.c_int, C.c_void_p
        L.hx_version.restype = C.c_char_p
        L.hx_last_error.restype = C.c_char_p
        L.hx_get_stream.restype = vp
        L.hx_set_stream.argtypes = [vp]
        L.hx_plan_create.restype = vp
        L.hx_plan_create.argtypes = [i, i, i]
        L.hx_plan_destroy.argtypes = [vp]
        L.hx_plan_destroy.restype = None
        L.hx_plan_scratch_bytes.argtypes = [vp]
        L.hx_plan_scratch_bytes.restype = C.c_int64
        L.hx_plan_release_scratch.argtypes = [vp]
        L.hx_set_scratch_budget.argtypes = [C.c_double]
        L.hx_plan_last_chunks.argtypes = [vp]
        L.hx_plan_mfma_flops.argtypes = [vp, i, i, C.POINTER(C.c_double)]
        L.hx_plan_executed_flops.argtypes = [vp, i, i, C.POINTER(C.c_double)]
        L.hx_executed_flops.argtypes = [C.POINTER(C.c_double), i]
        L.hx_measure_peaks.argtypes = [C.POINTER(C.c_double)]
        L.hx_measured_mfma_clock.restype = C.c_double
        L.hx_map2alm.argtypes = [vp, i, i, dp, dp, dp, dp, dp, i]
        L.hx_alm2map.argtypes = [vp, i, i, dp, dp]
        L.hx_map2alm_multi.argtypes = [vp, i, vp, vp, vp, vp, dp, dp, vp]
        L.hx_map2alm_list.argtypes = [vp, i, vp, vp, vp, dp, dp, dp, dp, i]
        L.hx_copy.argtypes = [vp, vp, C.c_int64]
        L.hx_alm2cl_pairs.argtypes = [i, vp, vp, i, i, vp, vp, dp]
        L.hx_alm2cl_pairs_range.argtypes = [i, vp, vp, i, i, vp, vp, i, i, i, dp]
        L.hx_gauss_legendre.argtypes = [i, dp, dp]
        L.hx_gauss_legendre_dd.argtypes = [i, dp, dp, dp]
        L.hx_wigner_d_table.argtypes = [i, i, i, i, dp, dp]
        L.hx_mixmat.argtypes = [dp, i, i, i, i, i, i, dp]
        L.hx_mixmat_eb.argtypes = [dp, i, i, i, i, dp]
        L.hx_mixmat_batch.argtypes = [i, dp, i, i, i, i, vp, vp, vp, vp]
        L.hx_mixctx_create.restype = vp
        L.hx_mixctx_create.argtypes = [i, i, i]
        L.hx_mixctx_apply.argtypes = [vp, dp, i, i, dp]
        L.hx_mixctx_destroy.argtypes = [vp]
        L.hx_mixctx_set_bins.argtypes = [vp, i, vp, dp, dp]
        L.hx_mixctx_apply_binned.argtypes = [vp, dp, i, i, dp]
        L.hx_mixctx_destroy.restype = None
        L.hx_cl2corr.argtypes = [i, i, dp, dp]
        L.hx_corr2cl.argtypes = [i, i, dp, dp]
        L.hx_ang2pix_ring.argtypes = [i, C.c_int64, dp, dp, dp]
        L.hx_map_values.argtypes = [i, C.c_int64, dp, dp, i, dp, dp, i]
        L.hx_ud_grade.argtypes = [i, i, i, dp, dp]
        L.hx_reorder.argtypes = [i, i, i, dp, dp]
        L.hx_matvec.argtypes = [i, i, dp, i, dp, dp]
        L.hx_pinv.argtypes = [i, i, dp, C.c_double, dp, C.POINTER(C.c_double)]
        L.hx_alm_resample.argtypes = [i, i, i, dp, dp]
        L.hx_region_maps.argtypes = [C.c_int64, i, dp, dp, C.c_double, dp]
        L.hx_alm_subtract.argtypes = [C.c_int64, dp, i, vp, dp]
        L.hx_fits_unpack_f64.argtypes = [C.c_int64, i, i, C.c_int64, C.c_int64, C.c_int64, dp, dp]
        L.hx_fits_pack_f64.argtypes = [C.c_int64, i, i, C.c_int64, C.c_int64, C.c_int64, dp, dp]
        L.hx_pointsht_create.argtypes = [i, C.c_double]
        L.hx_pointsht_create.restype = vp
        L.hx_pointsht_destroy.argtypes = [vp]
        L.hx_pointsht_destroy.restype = None
        L.hx_pointsht_info.argtypes = [vp, C.POINTER(C.c_int)]
        L.hx_pointsht_adjoint.argtypes = [vp, i, i, C.c_int64, dp, dp, dp]
        L.hx_ring_modes_size.argtypes = [vp, i]
        L.hx_ring_modes_size.restype = C.c_int64
        L.hx_ring_modes.argtypes = [vp, i, dp, dp, dp, i, vp, vp, i, vp]
        L.hx_legendre_from_modes.argtypes = [vp, i, i, vp, i, i, i, dp, dp]
        L.hx_allgather_alms.argtypes = [vp, i, vp, dp]
        L.hx_pixel_weights_size.argtypes = [i]
        L.hx_pixel_weights_size.restype = C.c_int64
        L.hx_pixel_weights_expand.argtypes = [i, C.c_int64, dp, dp]
        L.hx_host_alloc.argtypes = [C.c_int64, C.POINTER(C.c_void_p)]
        L.hx_host_free.argtypes = [vp]
        L.hx_mixmat_gemm_clock.restype = C.c_double
        L.hx_timer_stop.argtypes = [C.POINTER(C.c_float)]
        L.hx_profile_get.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        _lib = L
        return L


def check(rc: int):
    if rc != HX_OK:
        raise HxError(rc, load().hx_last_error().decode(errors="replace"))


def ptr(obj):
    """Raw address of a numpy array, a torch tensor (host or device), an int, or None."""
    if obj is None:
        return None
    if isinstance(obj, int):
        return C.c_void_p(obj)
    if isinstance(obj, np.ndarray):
        if not obj.flags.c_contiguous:
            raise ValueError("array must be C-contiguous")
        return C.c_void_p(obj.ctypes.data)
    if hasattr(obj, "data_ptr"):  # torch tensor
        if not obj.is_contiguous():
            raise ValueError("tensor must be contiguous")
        if getattr(obj, "is_cuda", False):
            # libhxsht launches on its own stream: whatever torch queued on this tensor
            # (fills, copies, collectives) has to be complete before the library touches it
            import torch

            torch.cuda.current_stream(obj.device).synchronize()
        return C.c_void_p(obj.data_ptr())
    raise TypeError(f"cannot take the address of {type(obj)!r}")


def init(device: int | None = None):
    L = load()
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) % max(L.hx_device_count(), 1)
    check(L.hx_init(int(device)))
    global _inited
    _inited = True


_inited = False


def ensure_init():
    """Initialise the library once on the device of this process (LOCAL_RANK aware)."""
    global _inited
    if not _inited:
        init(None)
        _inited = True


def device_count() -> int:
    return load().hx_device_count()


def measure_peaks():
    """{hbm_read_gbs, hbm_copy_gbs, fp64_mfma_tflops, fp64_valu_tflops} sustained by this device."""
    ensure_init()
    out = (C.c_double * 4)()
    check(load().hx_measure_peaks(out))
    return {"hbm_read_gbs": out[0], "hbm_copy_gbs": out[1], "fp64_mfma_tflops": out[2], "fp64_valu_tflops": out[3],
            "fp64_mfma_clock_ghz": load().hx_measured_mfma_clock()}


def synchronize():
    check(load().hx_synchronize())


def release_caches():
    """Hand back the HBM the library keeps between calls outside plans and contexts: the cache of ``mixmat`` / ``mixmat_eb`` (tables of
    the last sizes + host-staging buffer, ~3 GB at L = 6144) and the buffers of ``alm2cl_pairs`` (<= 512 MB): hx_release_caches."""
    check(load().hx_release_caches())


def copy(dst, src):
    """dst <- src (arrays or tensors of equal byte size, host or device) through the library's staging pipeline."""
    nb = int(src.nbytes) if hasattr(src, "nbytes") else int(src.numel() * src.element_size())
    nd = int(dst.nbytes) if hasattr(dst, "nbytes") else int(dst.numel() * dst.element_size())
    if nb != nd:
        raise ValueError(f"copy of {nb} bytes into {nd}")
    check(load().hx_copy(ptr(dst), ptr(src), nb))


def executed_flops(reset=False):
    """(matrix-instruction flops, vector-unit flops) the Legendre analysis kernels executed since the last reset, as counted by
    the kernels themselves."""
    ensure_init()
    out = (C.c_double * 2)()
    check(load().hx_executed_flops(out, 1 if reset else 0))
    return out[0], out[1]


def set_scratch_budget(nbytes: float):
    """HBM one analysis m-chunk may use for its operands and partial sums (0 = automatic)."""
    check(load().hx_set_scratch_budget(float(nbytes)))


class Timer:
    """HIP-event timer on the library stream."""

    def __enter__(self):
        check(load().hx_timer_start())
        self.ms = None
        return self

    def __exit__(self, *exc):
        t = C.c_float()
        check(load().hx_timer_stop(C.byref(t)))
        self.ms = float(t.value)
        return False


def profile_enable(on=True):
    check(load().hx_profile_enable(1 if on else 0))


def profile_reset():
    check(load().hx_profile_reset())


def profile_get(name: str):
    n = C.c_int()
    ms = C.c_double()
    check(load().hx_profile_get(name.encode(), C.byref(n), C.byref(ms)))
    return n.value, ms.value


# ---- host arrays of large results ------------------------------------------------------------------------------------
# A mixing-matrix build at L = 6144 returns 0.9 GB.  Into a FRESH numpy array (the default, as the reference's convolvecl call
# returns one) that costs 30-100 ms of first-touch page faults on top of the 16 ms the bytes need over PCIe -- more than the GPU spends
# on the matrices (19 ms).  A caller that builds many matrices in a loop and hands each one on (heracles.twopoint.mixing_matrices with
# an `out` mapping that writes them away, heracles/twopoint.py:393-397) passes `out=` instead: an array it owns and re-uses -- any
# C-contiguous float64 array of the right shape, or `pinned_empty(shape)`, page-locked memory the GPU reaches by DMA without the
# staging copy.  Ownership is the caller's and visible to it; the library keeps no pool of released results (round 4's _HostPool
# decided from reference counts whether a block could be handed out again: removed).
def pinned_empty(shape, dtype=None):
    """np.empty in page-locked host memory (hx_host_alloc); freed when the array and every view of it are gone."""
    import weakref

    dtype = np.dtype(np.float64 if dtype is None else dtype)
    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if nbytes == 0:
        return np.empty(shape, dtype)
    ensure_init()
    L = load()
    p = C.c_void_p()
    check(L.hx_host_alloc(C.c_int64(nbytes), C.byref(p)))
    raw = (C.c_char * nbytes).from_address(p.value)
    weakref.finalize(raw, L.hx_host_free, C.c_void_p(p.value))  # (numpy keeps `raw` alive as the base of every view)
    return np.frombuffer(raw, dtype=dtype).reshape(shape)


def result_array(shape, out=None):
    """The destination of a result: a fresh numpy array, or the caller's `out` (numpy array or anything with data_ptr / shape --
    a torch tensor, host or device -- of that shape, float64, C-contiguous), which is then what the call returns."""
    shape = tuple(int(x) for x in shape)
    if out is None:
        return np.empty(shape)
    if tuple(out.shape) != shape:
        raise ValueError(f"out has shape {tuple(out.shape)}, the result has {shape}")
    if isinstance(out, np.ndarray):
        if out.dtype != np.float64 or not out.flags.c_contiguous or not out.flags.writeable:
            raise ValueError("out must be a writeable C-contiguous float64 array")
    elif hasattr(out, "data_ptr"):
        import torch

        if out.dtype != torch.float64 or not out.is_contiguous():
            raise ValueError("out must be a contiguous float64 tensor")
    else:
        raise TypeError(f"out: cannot write into {type(out)!r}")
    return out
