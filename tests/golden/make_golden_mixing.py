"""Golden vectors for ``invert_mixing_matrix`` / ``apply_mixing_matrix`` from the reference's own functions.

Run ONCE in the build container (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden_mixing.py

``heracles.twopoint`` is importable through the bare-package shim of make_golden.py (numpy only; ``convolvecl`` is absent, so the input
matrices are synthetic, seeded arrays of the shapes ``mixing_matrices`` returns).  Only inputs and outputs are stored -- no reference source text.
"""

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import ref_modules  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    import types

    sys.modules.setdefault("convolvecl", types.ModuleType("convolvecl"))  # (imported at call time by mixing_matrices only; not used here)
    h = ref_modules()
    Result = h.result.Result
    rng = np.random.default_rng(404)
    out = {}

    def smooth(n, m, width):
        """a mixing-matrix-like array: band of the given width around the diagonal plus a small dense part"""
        i, j = np.arange(n)[:, None], np.arange(m)[None, :]
        return np.exp(-0.5 * ((i * (m - 1) / max(n - 1, 1) - j) / width) ** 2) + 1e-3 * rng.standard_normal((n, m))

    cases = {
        "square": (31, 31), "tall": (41, 21), "wide": (11, 21),
    }
    for name, (n, m) in cases.items():
        mats = {
            ("POS", "POS", 0, 0): Result(smooth(n, m, 2.0), spin=(0, 0), axis=-2, ell=np.arange(n)),
            ("POS", "SHE", 0, 1): Result(smooth(n, m, 3.0), spin=(0, 2), axis=-2, ell=np.arange(n)),
            ("SHE", "SHE", 1, 1): Result(np.array([smooth(n, m, 2.5), 0.1 * smooth(n, m, 4.0), smooth(n, m, 2.0)]), spin=(2, 2), axis=-2, ell=np.arange(n)),
        }
        rconds = {("POS", "POS", 0, 0): 1e-2, ("POS", "SHE", 0, 1): 1e-5, ("SHE", "SHE", 1, 1): 1e-3}
        inv = h.twopoint.invert_mixing_matrix(mats, rcond=rconds)
        cls = {
            ("POS", "POS", 0, 0): Result(rng.standard_normal(n), spin=(0, 0), axis=-1),
            ("POS", "SHE", 0, 1): Result(rng.standard_normal((2, n)), spin=(0, 2), axis=-1),
            ("SHE", "SHE", 1, 1): Result(rng.standard_normal((2, 2, n)), spin=(2, 2), axis=-1),
        }
        app = h.twopoint.apply_mixing_matrix(cls, inv)
        for key in mats:
            ks = "|".join(str(k) for k in key)
            out[f"{name}/M/{ks}"] = np.asarray(mats[key].array)
            out[f"{name}/rcond/{ks}"] = np.array(rconds[key])
            out[f"{name}/inv/{ks}"] = np.asarray(inv[key].array)
            out[f"{name}/inv_ell/{ks}"] = np.asarray(inv[key].ell)
            out[f"{name}/cl/{ks}"] = np.asarray(cls[key].array)
            out[f"{name}/applied/{ks}"] = np.asarray(app[key].array)
    # rank-deficient input, the reference's own test case (tests/test_twopoint.py:425-447): matrices of ones
    ones = {("A", "A", 0, 0): Result(np.ones((11, 21)), spin=(0, 0), axis=-2, ell=np.arange(11)),
            ("B", "B", 0, 0): Result(np.ones((3, 11, 21)), spin=(2, 2), axis=-2, ell=np.arange(11))}
    inv = h.twopoint.invert_mixing_matrix(ones, rcond=1e-4)
    for key in ones:
        ks = "|".join(str(k) for k in key)
        out[f"ones/inv/{ks}"] = np.asarray(inv[key].array)
    np.savez_compressed(os.path.join(OUT, "reference_mixing.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
