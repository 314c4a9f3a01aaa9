#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (tools/profile_round.sh) into profiles/<tag>_*."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = sys.argv[2] if len(sys.argv) > 2 else f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)


def short(name):
    name = name.split("(")[0]
    if "distribution_elementwise" in name:
        return "at::native::distribution_elementwise_grid_stride_kernel<normal> (torch.randn input fill)"
    return name.replace("void ", "")


# 1. kernel stats of the traced bench command
stats = glob.glob(f"{src}/trace/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(stats)))
with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
bench = open(f"{src}/bench_trace.json").read().strip().splitlines()[-1]
open(f"profiles/{tag}_bench_under_rocprof.json", "w").write(bench + "\n")

# 2. counters (smaller workload, one pass per counter group)
out = [f"# PMC summary ({tag}): per-dispatch averages, workload = bench.py --nside 2048 --lmax 3072 --nbins 4\n"]
for d in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write"):
    fs = glob.glob(f"{src}/{d}/*/*_counter_collection.csv")
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
    for r in csv.DictReader(open(glob.glob(f"{src}/{d}/*/*_kernel_trace.csv")[0])):
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out.append(f"\n## {d}\n")
    for k in sorted(agg):
        n = len(seen[k])
        vals = ", ".join(f"{c}={v / n:.4g}" for c, v in sorted(agg[k].items()))
        out.append(f"- `{k}` ({n} dispatches, avg {sum(dur[k]) / max(len(dur[k]), 1) / 1e6:.3f} ms): {vals}")
open(f"profiles/{tag}_pmc_summary.md", "w").write("\n".join(out) + "\n")
print("wrote profiles/", tag)
