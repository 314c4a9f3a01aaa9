#!/bin/bash
# Pipelined Legendre kernel at full size with phases removed (tools/build_diag.sh 1 2 4 6 first)
for spec in "2 10" "0 10" "2 16"; do
  set -- $spec
  for n in 0 1 2 4 6; do
    lib=heracles_amd/libhxsht.so; [ $n != 0 ] && lib=tools/bin/libhxsht_abl$n.so
    HX_LIBRARY=$PWD/$lib NSIDE=${NSIDE:-4096} LMAX=${LMAX:-6144} SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>/dev/null | sed "s/^/abl $n ncomp $2: /"
  done
done
