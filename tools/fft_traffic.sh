#!/bin/bash
# HBM / L2 traffic of the ring Fourier kernels of tools/time_fft.py (8 spin-0 maps, nside 4096): one counter group per pass.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/fft_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 $REPO/tools/time_fft.py > $OUT/$tag.log 2>&1
done
cd $REPO && python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/fft_pmc/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            if "subdft" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
        for k, v in agg.items():
            print("%-16s total over %3d dispatches (2 calls of 8 maps): %.4g  -> per call %.4g" % (k, len(disp[k]), v, v / 2))
PY
