cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export NSIDE=4096 LMAX=6144 HX_PIPE_ONESET=0 HX_LEG_KERNEL=duo
(
for spec in "2 8" "0 8"; do set -- $spec
for t in default duoabl1 duoabl2 duoabl4 duoabl8 duoabl5 duoabl6; do
lib=""; [ "$t" != default ] && lib=$PWD/tools/bin/libhxsht_$t.so
HX_LIBRARY=$lib SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>/dev/null | sed "s|^|$t: |"
done; done
) > gpurun_out/r4_t2_abl.log 2>&1
cat gpurun_out/r4_t2_abl.log
