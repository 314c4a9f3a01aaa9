cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export NSIDE=4096 LMAX=6144
(
for spec in "2 6" "2 8" "2 10" "2 12" "2 14" "2 16" "2 18" "2 20" "0 5" "0 6" "0 8" "0 9" "0 10" "0 12" "0 16"; do set -- $spec
SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>&1 | grep -v amdgpu.ids | sed "s|^|duo:  |"
done
) > gpurun_out/r4_t6_shapes.log 2>&1
cat gpurun_out/r4_t6_shapes.log
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r4_t6_tests.log 2>&1
tail -5 gpurun_out/r4_t6_tests.log
