#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (tools/profile_round.sh) into profiles/<tag>_*."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = sys.argv[2] if len(sys.argv) > 2 else f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)


def short(name):
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    if "distribution_elementwise" in name:
        return "at::native::distribution_elementwise_grid_stride_kernel<normal> (torch.randn input fill)"
    return name.replace("void ", "")


# 1. kernel stats of the traced bench command
stats = max(glob.glob(f"{src}/trace/*/*_kernel_stats.csv"), key=os.path.getmtime)  # a stale run directory may sit beside the new one
rows = list(csv.DictReader(open(stats)))
with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
bench = open(f"{src}/bench_trace.json").read().strip().splitlines()[-1]
open(f"profiles/{tag}_bench_under_rocprof.json", "w").write(bench + "\n")

# 2. counters (smaller workload, one pass per counter group)
out = [f"# PMC summary ({tag}): per-dispatch averages at FULL size (bench.py --steps 1 --warmup 0: nside 4096, lmax 6144, 10 + 10 maps)\n",
       "clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x clock)\n"]
for d in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write"):
    fs = sorted(glob.glob(f"{src}/{d}/*/*_counter_collection.csv"), key=os.path.getmtime, reverse=True)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        if not k.startswith("hx::"):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0].replace("_counter_collection.csv", "_kernel_trace.csv"))):
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out.append(f"\n## {d}\n")
    for k in sorted(agg):
        n = len(seen[k])
        vals = ", ".join(f"{c}={v / n:.4g}" for c, v in sorted(agg[k].items()))
        avg_ms = sum(dur[k]) / max(len(dur[k]), 1) / 1e6
        extra = ""
        if "GRBM_GUI_ACTIVE" in agg[k] and avg_ms > 0:
            ghz = agg[k]["GRBM_GUI_ACTIVE"] / n / 8 / (avg_ms * 1e-3) / 1e9
            extra = f" -> clock {ghz:.2f} GHz"
            if agg[k].get("SQ_VALU_MFMA_BUSY_CYCLES"):
                busy = agg[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (1024 * avg_ms * 1e-3 * ghz * 1e9)
                tf = agg[k]["SQ_INSTS_VALU_MFMA_F64"] / n
                extra += f", MFMA pipe busy {100 * busy:.1f} % of the kernel"
        out.append(f"- `{k}` ({n} dispatches, avg {avg_ms:.3f} ms): {vals}{extra}")
open(f"profiles/{tag}_pmc_summary.md", "w").write("\n".join(out) + "\n")
# 3. HBM traffic per launch (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction of MI355X_MICROARCH.md)
res = collections.defaultdict(dict)
for kind, d in (("fetch", "pmc_fetch"), ("write", "pmc_write")):
    fs = sorted(glob.glob(f"{src}/{d}/*/*_counter_collection.csv"), key=os.path.getmtime, reverse=True)
    if not fs:
        continue
    agg, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        k = short(r["Kernel_Name"])
        if k.startswith("hx::k_"):
            agg[k] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
    for k in agg:
        res[k][kind] = (agg[k], len(n[k]))
tr = {}
for k, v in sorted(res.items()):
    fk, nf = v.get("fetch", (0, 1))
    wk, nw = v.get("write", (0, 1))
    tr[k] = {"launches": nf, "fetch_size_kb_per_launch": fk / nf, "write_size_kb_per_launch": wk / nw,
             "hbm_bytes_per_launch": (2 * fk / nf + wk / nw) * 1024}
if tr:
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes, tools/profile_round.sh) of "
                         "`bench.py --steps 1 --warmup 0` at nside 4096 / lmax 6144; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB: "
                         "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950", "kernels": tr},
              open(f"profiles/{tag}_traffic.json", "w"), indent=1)
print("wrote profiles/", tag)
