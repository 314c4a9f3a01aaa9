#!/bin/bash
# PMC pass over the batched synthesis at the bench size (tools/time_synth_duo.py): matrix pipe busy, instruction mix, LDS conflicts.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_synth
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CASES=${CASES:-2:10,0:10}
i=0
for set in "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $REPO/tools/time_synth_duo.py > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "k_synth_duo" not in k and "k_synth_spectrum" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen[k].add(r["Dispatch_Id"])
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f.replace("_counter_collection.csv", "_kernel_trace.csv"))):
            dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, c in agg.items():
            n = len(seen[k]); ms = sum(dur[k]) / max(len(dur[k]), 1) / 1e6
            line = f"{k}: {n} dispatches, {ms:.1f} ms each;"
            for cn, v in sorted(c.items()): line += f" {cn}={v / n:.4g}"
            if "GRBM_GUI_ACTIVE" in c:
                clk = c["GRBM_GUI_ACTIVE"] / n / 8 / (ms * 1e-3) / 1e9
                busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (1024 * ms * 1e-3 * clk * 1e9)
                line += f" | clock {clk:.2f} GHz, matrix pipe busy {busy:.3f}, matrix flops {c['SQ_INSTS_VALU_MFMA_F64'] / n:.4g} wave-instructions"
            print(line)
PY
