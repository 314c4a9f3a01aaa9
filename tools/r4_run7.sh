cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export NSIDE=4096 LMAX=6144
timeout -k 10 900 python -m pytest tests/test_gpu_sht.py tests/test_gpu_fullsize.py -q -m gpu -x > gpurun_out/r4_t7_tests.log 2>&1
tail -3 gpurun_out/r4_t7_tests.log
(
for rep in 1 2; do
for spec in "2 20" "2 10" "2 8" "0 10" "0 8" "0 16"; do set -- $spec
SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>&1 | grep -v amdgpu.ids | sed "s|^|duo:  |"
done; done
) > gpurun_out/r4_t7_duo.log 2>&1
cat gpurun_out/r4_t7_duo.log
