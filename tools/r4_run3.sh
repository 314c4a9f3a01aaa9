cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export NSIDE=4096 LMAX=6144 HX_LEG_KERNEL=duo
(
for spec in "2 18" "2 8"; do set -- $spec
for t in duostamp duostamp1; do
lib=$PWD/tools/bin/libhxsht_$t.so
HX_LIBRARY=$lib SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>&1 | grep -v amdgpu.ids | tail -9 | sed "s|^|$t: |"
done; done
) > gpurun_out/r4_t3_stamp.log 2>&1
cat gpurun_out/r4_t3_stamp.log
