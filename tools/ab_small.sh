#!/bin/bash
# Small batches (<= 4 spin-0 maps / <= 2 spin-2 fields: one sweep per map / field of the vector-unit kernel) at full size on one
# device; HX_LIBRARY selects another build.  (Round 1's 4x4x4 kernels, retired in round 3, took 61 / 61 / 91 ms for 1 / 2 / 4
# spin-0 maps and 108 / 142 ms for 1 / 2 spin-2 fields.)
for spec in "0 1" "2 2" "0 2" "0 4" "2 4"; do set -- $spec
env NSIDE=${NSIDE:-4096} LMAX=${LMAX:-6144} SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>/dev/null
done
