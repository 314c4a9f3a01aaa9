"""Timeline of a stamped mixing-matrix GEMM launch (HX_GEMM_STAMP_FILE of a -DHX_GEMM_STAMP build).  usage: analyse_gemm_stamps.py FILE"""
import sys
import numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.uint64)
d = d[d[:, 1] > 0]
t0 = d[:, 1].min()
st = (d[:, 1] - t0).astype(float) / 100e6 * 1e3
en = (d[:, 2] - t0).astype(float) / 100e6 * 1e3
hw = d[:, 4].astype(int); xcc = d[:, 3].astype(int)
cu = (xcc << 16) | (hw & 0xff00)    # (xcc, se, sh, cu) of HW_ID bits 8-15
dur = en - st
print("groups", len(d), "launch %.2f ms" % en.max(), "tile: mean %.3f min %.3f max %.3f ms" % (dur.mean(), dur.min(), dur.max()))
# alone or paired: another group on the same CU for more than half of its life
paired = np.zeros(len(d), bool)
for i in range(len(d)):
    o = (cu == cu[i]); o[i] = False
    ov = np.clip(np.minimum(en[o], en[i]) - np.maximum(st[o], st[i]), 0, None).sum()
    paired[i] = ov > 0.5 * dur[i]
print("paired tiles", paired.sum(), "mean %.3f ms;  alone" % dur[paired].mean(), (~paired).sum(), "mean %.3f ms" % (dur[~paired].mean() if (~paired).any() else 0))
ts = np.linspace(0, en.max(), 31)[:-1]
print("running groups at 30 times:", [int(((st <= t) & (en > t)).sum()) for t in ts])
