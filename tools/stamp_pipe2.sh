#!/bin/bash
# which part of the recursion step costs what: 8 = stamps, +16 no tile store, +32 no coefficient read, +64 no promote, +128 no chain dependency
for n in 8 24 40 72 136 248; do
  echo "== abl $n"
  HX_LIBRARY=$PWD/tools/bin/libhxsht_abl$n.so NSIDE=4096 LMAX=6144 SPIN=${SPIN:-2} NCOMP=${NCOMP:-10} python tools/leg_only.py 2>&1 | grep -E "mfma\|\|rec|rec alone|dead alone|mfma\|\|dead|legendre ms" | tail -5
done
