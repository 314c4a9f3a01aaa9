"""The C-ABI library loads, exports every symbol include/hxsht.h declares, and fails
loudly (no CPU fallback) when there is no GPU.  No compute calls here."""

import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "hxsht.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    import heracles_amd

    lib = ctypes.CDLL(heracles_amd._lib.library_path())
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/hxsht.h but not exported"
    assert sorted(heracles_amd._lib.SYMBOLS) == syms


def test_version_and_device_count():
    import heracles_amd

    L = heracles_amd._lib.load()
    assert b"gfx950" in L.hx_version()
    assert L.hx_device_count() >= 0


def test_no_cpu_fallback_without_gpu():
    import numpy as np

    import heracles_amd

    if heracles_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(heracles_amd.HxError) as e:
        heracles_amd.alm2cl(np.zeros(6, complex))
    assert e.value.code == -2
    with pytest.raises(heracles_amd.HxError):
        heracles_amd.mixmat(np.ones(4))
    with pytest.raises(heracles_amd.HxError):
        heracles_amd.Plan(4, 4)
    m = heracles_amd.HipHealpixMapper(4, 4, deconvolve=False)
    with pytest.raises(heracles_amd.HxError):
        m.transform(m.create(), spin=0)
    # round 5's entry points: page-locked memory, the mixing-matrix cache, double-double nodes
    with pytest.raises(heracles_amd.HxError):
        heracles_amd.pinned_empty((4, 4))
    with pytest.raises(heracles_amd.HxError):
        heracles_amd.mixmat_eb(np.ones(4), out=np.empty((3, 4, 4)))
    # round 6's entry points: binned mixing matrices (the context cannot even be created), the driver with bins
    import types

    with pytest.raises(heracles_amd.HxError):
        heracles_amd.MixmatContext(8, 8, 8)
    with pytest.raises(heracles_amd.HxError):
        heracles_amd.mixing_matrices({"P": types.SimpleNamespace(mask="V", spin=0)}, {("V", "V", 0, 0): np.ones(9)}, bins=np.array([0, 4, 9]))
    heracles_amd.release_caches()  # (nothing is held: not an error without a device)
    with pytest.raises(heracles_amd.HxError):
        heracles_amd.MixmatContext(3, 3, 3)
    x = np.empty(5)
    assert heracles_amd._lib.load().hx_gauss_legendre_dd(5, heracles_amd._lib.ptr(x), heracles_amd._lib.ptr(x.copy()), heracles_amd._lib.ptr(x.copy())) == -2
    assert heracles_amd._lib.load().hx_mixmat_gemm_clock() == 0.0


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "heracles_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "hxoracle" not in txt and "hx_oracle" not in txt and "oracle/" not in txt, f


def test_fft_core_host_emulation(tmp_path):
    exe = tmp_path / "t_fft"
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "csrc", "test_fft_core.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
