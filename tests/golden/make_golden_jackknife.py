"""Golden vectors for the "Full" footprint correction of the jackknife loop (heracles/dices/jackknife.py:411-450).

Run ONCE in the build container (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden_jackknife.py

`heracles.dices.jackknife` itself cannot be imported here: its module header pulls in `..mapping` (needs `coroutines`)
and `..io` (needs `fitsio`), both absent -- ordinary ModuleNotFoundErrors.  `_mask_correlation_ratio` and
`correct_footprint_naturalspice` are ten lines of glue around functions that ARE importable through the bare-package
shim of make_golden.py (`transforms.cl2corr` / `corr2cl`, `unmixing._naturalspice`, `result.binned`): the glue is
re-assembled below from those callees, in the order of jackknife.py:411-450, so every number in the file comes out
of the reference's own arithmetic.  Only inputs and outputs are stored.
"""

import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import key_str, ref_modules  # noqa: E402

try:
    from copy import replace
except ImportError:
    from dataclasses import replace

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    h = ref_modules()
    Result, binned = h.result.Result, h.result.binned
    cl2corr, corr2cl, _naturalspice = h.transforms.cl2corr, h.transforms.corr2cl, h.unmixing._naturalspice
    rng = np.random.default_rng(51)
    out = {}
    L, Lm = 24, 48
    ell, ellm = np.arange(L + 1), np.arange(Lm + 1)
    spins = {("POS", "POS", 0, 0): (0, 0), ("POS", "SHE", 0, 0): (0, 2), ("SHE", "SHE", 0, 0): (2, 2)}
    shapes = {("POS", "POS", 0, 0): (), ("POS", "SHE", 0, 0): (2,), ("SHE", "SHE", 0, 0): (2, 2)}
    cls = {}
    for k, shp in shapes.items():
        arr = rng.standard_normal(shp + (L + 1,)) / (1 + ell) ** 2
        if spins[k][0] or spins[k][1]:
            arr[..., :2] = 0.0
        cls[k] = Result(arr, spin=spins[k], axis=-1, ell=ell)
        out[f"cls/{key_str(k)}"] = arr
    mls0, mljk = {}, {}
    for k in (("VIS", "VIS", 0, 0), ("VIS", "WHT", 0, 0), ("WHT", "WHT", 0, 0)):
        base = 4 * np.pi * 0.4 * np.exp(-ellm * (ellm + 1) / 200.0) + 1e-4 / (1 + ellm) ** 2
        a0 = base * rng.uniform(0.9, 1.1)
        ajk = 0.8 * a0 * (1.0 + 0.05 * np.cos(ellm / 7.0))
        mls0[k] = Result(a0, spin=(0, 0), axis=-1, ell=ellm)
        mljk[k] = Result(ajk, spin=(0, 0), axis=-1, ell=ellm)
        out[f"mls0/{key_str(k)}"], out[f"mljk/{key_str(k)}"] = a0, ajk
    fields = {"POS": types.SimpleNamespace(mask="VIS", spin=0), "SHE": types.SimpleNamespace(mask="WHT", spin=2)}
    for tag, unmixed in (("mixed", False), ("unmixed", True)):
        # jackknife.py:411-423 (_mask_correlation_ratio)
        wmls0, wmljk = cl2corr(mls0), cl2corr(mljk)
        alphas = {}
        for key in list(wmljk.keys()):
            alpha = wmljk[key].array
            if not unmixed:
                alpha = alpha / wmls0[key].array
            alphas[key] = replace(mls0[key], array=alpha)
        for k, v in alphas.items():
            out[f"{tag}/alpha/{key_str(k)}"] = np.array(v.array)
        # jackknife.py:426-450 (correct_footprint_naturalspice)
        first_cls, first_mls = list(cls.values())[0], list(mls0.values())[0]
        lmax, lmax_mask = first_cls.shape[first_cls.axis[0]], first_mls.shape[first_mls.axis[0]]
        c = binned(cls, np.arange(0, lmax_mask + 1))
        wcls = _naturalspice(cl2corr(c), alphas, fields)
        res = binned(corr2cl(wcls), np.arange(0, lmax + 1))
        for k, v in res.items():
            out[f"{tag}/out/{key_str(k)}"] = np.asarray(v.array)
    np.savez_compressed(os.path.join(OUT, "reference_jackknife.npz"), **out)
    print(f"wrote {len(out)} arrays to reference_jackknife.npz")


if __name__ == "__main__":
    main()
