#!/bin/bash
# A/B of the small-batch Legendre path on one device: HX_VALU=1 (vector-unit kernel) vs 0 (round-1 4x4x4 kernels), full size
for spec in "0 1" "2 2" "0 2" "0 4" "2 4"; do set -- $spec
for v in 1 0; do
env HX_VALU=$v NSIDE=${NSIDE:-4096} LMAX=${LMAX:-6144} SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>/dev/null | sed "s|^|HX_VALU=$v: |"
done; done
