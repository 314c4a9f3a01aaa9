#!/bin/bash
for sp in 0 2; do for a in 0 1 2; do
  HX_ABLATE=$a SPIN=$sp python tools/leg_only.py 2>/dev/null | sed "s/^/ablate $a /"
done; done
