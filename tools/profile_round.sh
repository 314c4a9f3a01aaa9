#!/bin/bash
# Profiles of the bench command for profiles/ (run on the GPU box through gpurun); all passes at FULL size
# (nside 4096, lmax 6144, 10 + 10 maps).  usage: tools/profile_round.sh <tag>     then: tools/summarize_profile.py <tag>
set -o pipefail
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-host-leg"
# 1. kernel trace + stats of the default bench command (no CPU-baseline / host legs: they are host code)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 2 --warmup 1 $LEAN > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace done"
# 2. counters, each group in its own pass, one step, no warm-up (counters serialise kernels).  The FP64 probe of
#    hx_measure_peaks runs at the end of bench.py, so its MFMA loop appears in the same tables.
ONE="--steps 1 --warmup 0 --no-verify --no-single --no-niter3 $LEAN"   # (the mixing-matrix build stays in: k_mixmat_gemm is the one kernel the north star names as MFMA)
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $ONE > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
echo "sq done"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_lds -- python3 $REPO/bench.py $ONE > $OUT/pmc_lds.json 2> $OUT/pmc_lds.err
echo "lds done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ONE > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ONE > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "write done"
cd $OUT && find . -name "*.csv" | head -50 && du -sh .
