#!/bin/bash
# diagnostic: Legendre analysis kernel time at full size with phases removed (results wrong by design)
# ablate bits: 1 skip MFMA, 2 skip recursion (+MFMA), 8 count dead / live / mixed wave-blocks
for spin in 0 2; do
  for a in 0 1 2; do
    NSIDE=4096 LMAX=6144 SPIN=$spin HX_ABLATE=$a python tools/leg_only.py 2>/dev/null | sed "s/^/ablate $a: /"
  done
done
