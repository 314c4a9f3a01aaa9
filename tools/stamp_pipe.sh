#!/bin/bash
# cycle accounting of the pipelined Legendre kernel (tools/build_diag.sh 8 first)
for spec in "2 10" "0 10" "2 16"; do
  set -- $spec
  echo "== spin $1 ncomp $2"
  HX_LIBRARY=$PWD/tools/bin/libhxsht_abl8.so NSIDE=${NSIDE:-4096} LMAX=${LMAX:-6144} SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>&1 | grep -v amdgpu.ids | tail -9
done
for n in 4 6; do
  HX_LIBRARY=$PWD/tools/bin/libhxsht_abl$n.so NSIDE=4096 LMAX=6144 SPIN=2 NCOMP=10 python tools/leg_only.py 2>/dev/null | sed "s/^/abl $n ncomp 10: /"
done
