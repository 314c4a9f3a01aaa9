#!/bin/bash
# hx_pinv: inner Jacobi sweeps per Gram matrix against outer sweeps and seconds (run on the GPU box)
for inner in 12 4 2 1; do
  echo "inner sweeps <= $inner"
  HX_SVD_INNER=$inner SIZES=2049,4097 NREF=0 timeout -k 10 120 python tools/time_pinv.py 2>&1 | tail -2
done
