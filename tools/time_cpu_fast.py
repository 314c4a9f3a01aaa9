"""Threads vs time of the vectorised CPU baseline on this host (cgroup quota, affinity)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import hxfast as hf
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "quota", hf.cpu_quota(), "omp default", hf.num_threads(), flush=True)
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/proc/loadavg"):
    try: print(f, open(f).read().strip())
    except OSError as e: print(f, e)
nside, lmax = int(os.environ.get("NSIDE", 2048)), int(os.environ.get("LMAX", 3072))
rng = np.random.default_rng(1)
x = rng.standard_normal((1, 12 * nside * nside))
F0 = 8.0 * 2 * nside * (lmax + 1) * (lmax + 2) // 2
for nt in [int(v) for v in os.environ.get("THREADS", "8,16,32,64,128").split(",")]:
    hf.set_threads(nt)
    t = time.time(); _, tim = hf.map2alm(x, nside, lmax, spin=0); dt = time.time() - t
    print(f"threads {nt}: wall {dt:.2f}s ring {tim[0]:.2f}s legendre {tim[1]:.2f}s -> {F0/tim[1]/1e9:.0f} GF/s algorithmic", flush=True)
