#!/bin/bash
# A/B of the builds of tools/build_valu_variants.sh on one device: one spin-0 map and one spin-2 field at full size
for tag in "$@"; do
for spec in "0 1" "2 2"; do set -- $spec
env HX_LIBRARY=$PWD/tools/bin/libhxsht_$tag.so NSIDE=${NSIDE:-4096} LMAX=${LMAX:-6144} SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>/dev/null | sed "s|^|$tag: |"
done; done
