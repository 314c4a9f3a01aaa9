#!/bin/bash
# A/B of two builds on one device with tools/leg_only.py at full size: tools/ab_lib.sh <tagA|default> <tagB|default> ["spin ncomp" ...]
a=$1; b=$2; shift; shift
[ $# -eq 0 ] && set -- "2 20" "0 10"
for rep in 1 2; do
for spec in "$@"; do set -- $spec
for t in $a $b; do
lib=""; [ "$t" != default ] && lib=$PWD/tools/bin/libhxsht_$t.so
env HX_LIBRARY=$lib NSIDE=${NSIDE:-4096} LMAX=${LMAX:-6144} SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>/dev/null | sed "s|^|$t: |"
done; done; done
