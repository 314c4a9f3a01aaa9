cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export NSIDE=4096 LMAX=6144
(
for rep in 1 2; do
for spec in "2 20" "2 18" "2 10" "2 8" "0 10" "0 8"; do set -- $spec
SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>&1 | grep -v amdgpu.ids | sed "s|^|pipe: |"
HX_LEG_KERNEL=duo SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>&1 | grep -v amdgpu.ids | sed "s|^|duo:  |"
done; done
) > gpurun_out/r4_t4_duo.log 2>&1
HX_LEG_KERNEL=duo timeout -k 10 900 python -m pytest tests/test_gpu_sht.py -q -m gpu > gpurun_out/r4_t4_tests.log 2>&1
tail -5 gpurun_out/r4_t4_tests.log
cat gpurun_out/r4_t4_duo.log
