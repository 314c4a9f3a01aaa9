#!/bin/bash
# diagnostic: time the Legendre kernel with phases removed (results are wrong by design)
for a in 0 1 2 3 4 5 7; do
  HX_ABLATE=$a python bench.py --nside 2048 --lmax 3072 --nbins 4 --steps 2 --warmup 1 --no-cpu-baseline --no-mixmat 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate $a', 'legendre ms/step', round(d['kernels']['legendre_analysis']['ms_per_step'],2))"
done
