"""Timeline of the work-groups of a stamped launch (HX_DUO_STAMP_FILE of a -DHX_DUO_ABL=128 build): occupancy over time, tail,
per-XCD finish times.  usage: analyse_stamps.py FILE"""
import sys
import numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.uint64)
d = d[d[:, 2] > 0]
t0 = d[:, 1].min()
st = (d[:, 1] - t0).astype(float) / 100e6 * 1e3
en = (d[:, 2] - t0).astype(float) / 100e6 * 1e3
m = d[:, 0].astype(int); xcc = d[:, 3].astype(int); hw = d[:, 4].astype(int); ngr = d[:, 5].astype(int)
dur = en - st
T = en.max()
print("groups", len(d), "kernel %.1f ms" % T, "sum of durations / 512 slots = %.1f ms" % (dur.sum() / 512))
ts = np.linspace(0, T, 41)[:-1]
print("running groups at 40 times:", [int(((st <= t) & (en > t)).sum()) for t in ts])
for x in range(8):
    s = xcc == x
    print(" xcc", x, "groups", s.sum(), "busy sum %.0f ms" % dur[s].sum(), "last end %.1f" % en[s].max(), "first idle slot (64th-last end) %.1f" % np.sort(en[s])[-64])
print("longest groups: m, ring groups, ms:", [(int(m[i]), int(ngr[i]), round(dur[i], 1)) for i in np.argsort(-dur)[:6]])
print("start of m=0..3:", st[np.argsort(m)][:4].round(2), " starts of the last 8 groups:", np.sort(st)[-8:].round(1))
order = np.argsort(m)
print("duration by m (every 512th):", [(int(m[i]), round(dur[i], 1)) for i in order[::512]])
