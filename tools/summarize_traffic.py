#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>/ (tools/pmc_traffic.sh) -> profiles/<tag>_traffic.json: HBM bytes per launch."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def short(name):
    return name.split("(")[0].replace("void ", "")


res = collections.defaultdict(dict)
for kind in ("fetch", "write"):
    # a stale run directory may sit beside the new one: take the newest
    f = max(glob.glob(f"gpurun_out/pmc_{tag}/{kind}/*/*counter_collection.csv"), key=os.path.getmtime)
    agg, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k.startswith("hx::k_"):
            agg[k] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
    for k in agg:
        res[k][kind] = (agg[k], len(n[k]))
out = {}
for k, v in sorted(res.items()):
    fk, nf = v.get("fetch", (0, 1))
    wk, nw = v.get("write", (0, 1))
    out[k] = {"launches": nf, "fetch_size_kb_per_launch": fk / nf, "write_size_kb_per_launch": wk / nw,
              "hbm_bytes_per_launch": (2 * fk / nf + wk / nw) * 1024}
    print(f"{k:40s} launches {nf:3d}  fetch(x2) {2 * fk / nf / 1e6:9.2f} GB  write {wk / nw / 1e6:9.2f} GB")
json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes, tools/pmc_traffic.sh) of "
                     "`bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-mixmat` at nside 4096 / lmax 6144; bytes = "
                     "(2 x FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950",
           "kernels": out}, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
