#!/bin/bash
# Profiles of the bench command for profiles/ (run on the GPU box through gpurun).
# usage: tools/profile_round.sh <tag>
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the default bench command (no CPU baseline leg: it is host code)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
# 2. counters, each group in its own pass (smaller workload: counters serialise kernels)
SMALL="--nside 2048 --lmax 3072 --nbins 4 --steps 1 --warmup 1 --no-cpu-baseline --no-mixmat"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $SMALL > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -- python3 $REPO/bench.py $SMALL > $OUT/pmc_lds.json 2> $OUT/pmc_lds.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $SMALL > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $SMALL > $OUT/pmc_write.json 2> $OUT/pmc_write.err
cd $OUT && find . -name "*.csv" | head -50 && du -sh .
