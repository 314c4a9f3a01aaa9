cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export NSIDE=4096 LMAX=6144
HX_LEG_KERNEL=duo timeout -k 10 900 python -m pytest tests/test_gpu_sht.py -q -m gpu -x > gpurun_out/r4_t5_tests.log 2>&1
tail -3 gpurun_out/r4_t5_tests.log
(
for spec in "2 20" "2 18" "2 10" "2 8" "0 10" "0 8"; do set -- $spec
SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>&1 | grep -v amdgpu.ids | sed "s|^|pipe: |"
HX_LEG_KERNEL=duo SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>&1 | grep -v amdgpu.ids | sed "s|^|duo:  |"
done
export HX_LEG_KERNEL=duo
for spec in "2 20" "2 8" "0 10"; do set -- $spec
for t in duostamp duostamp1; do
lib=$PWD/tools/bin/libhxsht_$t.so
HX_LIBRARY=$lib SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>&1 | grep -v amdgpu.ids | tail -9 | sed "s|^|$t: |"
done; done
) > gpurun_out/r4_t5_duo.log 2>&1
cat gpurun_out/r4_t5_duo.log
