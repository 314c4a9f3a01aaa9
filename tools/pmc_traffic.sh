#!/bin/bash
# HBM traffic of the bench's kernels at FULL size (nside 4096, lmax 6144): FETCH_SIZE and WRITE_SIZE
# in separate passes (they do not fit one pass), one timed step, no warm-up.
# usage (on the GPU box): tools/pmc_traffic.sh <tag>
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-mixmat"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/bench.py $ARGS > $OUT/write.json 2> $OUT/write.err
cd $OUT && find . -name "*counter_collection.csv" && du -sh .
