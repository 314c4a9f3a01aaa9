#!/bin/bash
# A/B of one environment knob on one device: tools/ab_env.sh VAR valueA valueB  (pipelined Legendre kernel, full size)
var=$1; a=$2; b=$3
for rep in 1 2; do
for spec in "2 10" "0 10"; do set -- $spec
for v in $a $b; do
env $var=$v NSIDE=${NSIDE:-4096} LMAX=${LMAX:-6144} SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>/dev/null | sed "s|^|$var=$v: |"
done; done; done
