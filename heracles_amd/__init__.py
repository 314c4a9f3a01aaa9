"""heracles_amd -- MI355X-native backend for the harmonic-space two-point hot path of
Heracles (map2alm/alm2map, all-pairs alm x alm -> Cl, Wigner-3j mixing matrices).

Everything numerical runs in libhxsht.so (hand-written HIP for gfx950, C ABI in
include/hxsht.h).  There is no CPU fallback.
"""

from . import _lib
from ._lib import HxError, device_count, init, pinned_empty, release_caches, synchronize
from .binning import BinPlan, binned
from .core import DeviceArray, Result, TocDict, toc_match, update_metadata
from .discrete import HipDiscreteMapper, PointSHT, alm_resample, get_point_sht
from .jackknife import RegionAlms, jackknife_cls, region_alms
from .mapper import HipHealpixMapper
from .mapping import transform
from .sht import Plan, get_plan
from .transforms import cl2corr, corr2cl, gauss_legendre, wigner_d_table
from .fits import read_vmap
from .twopoint import (
    alm2cl,
    alm2cl_pairs,
    alm2lmax,
    angular_power_spectra,
    apply_mixing_matrix,
    debias_cls,
    MixmatContext,
    invert_mixing_matrix,
    mixing_matrices,
    mixmat,
    mixmat_eb,
    mixmat_release,
    split_requests,
)
from .unmixing import naturalspice

__all__ = [
    "HipHealpixMapper", "HipDiscreteMapper", "PointSHT", "get_point_sht", "alm_resample", "Plan", "get_plan", "HxError", "init", "device_count", "synchronize",
    "alm2cl", "alm2cl_pairs", "alm2lmax", "angular_power_spectra", "debias_cls",
    "mixing_matrices", "mixmat", "mixmat_eb", "cl2corr", "corr2cl", "gauss_legendre",
    "wigner_d_table", "naturalspice", "Result", "TocDict", "toc_match", "update_metadata", "DeviceArray",
    "pinned_empty", "release_caches", "mixmat_release", "split_requests", "binned", "BinPlan", "MixmatContext", "jackknife_cls", "region_alms", "RegionAlms", "transform", "read_vmap", "apply_mixing_matrix", "invert_mixing_matrix",
]
