#!/bin/sh
# Registers, spills, scratch and LDS of every kernel of one source file (cross-compiles, no GPU needed):
#   tools/kernel_resources.sh heracles_amd/csrc/hx_analysis.hip [pattern] [extra hipcc flags]
SRC=$1; PAT=${2:-.}; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/heracles_amd/csrc -I$ROOT/include "$@" \
    -Rpass-analysis=kernel-resource-usage -c "$SRC" -o /dev/null 2>&1 |
python3 -c '
import re, subprocess, sys
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\d+)", line)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
pat = re.compile(sys.argv[1])
for k, v in rows.items():
    if pat.search(k):
        print("%-60s vgpr %3d agpr %3d spill %3d scratch %4d lds %6d occ %s" % (k[-60:], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("VGPRs Spill", -1), v.get("ScratchSize", -1), v.get("LDS Size", -1), v.get("Occupancy", "?")))
' "$PAT"
