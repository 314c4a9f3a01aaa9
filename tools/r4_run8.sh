cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export NSIDE=4096 LMAX=6144
(
for rep in 1 2; do
for spec in "2 20" "2 10" "0 10"; do set -- $spec
for t in default nofl nofm nofmfl; do
lib=""; [ "$t" != default ] && lib=$PWD/tools/bin/libhxsht_$t.so
HX_LIBRARY=$lib SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>&1 | grep -v amdgpu.ids | sed "s|^|$t: |"
done; done; done
) > gpurun_out/r4_t8_ab.log 2>&1
cat gpurun_out/r4_t8_ab.log
