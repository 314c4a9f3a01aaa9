# same-device A/B of HX_DUO_GAP builds (tools/build_variant.sh hx_analysis.hip g<N> "-DHX_DUO_GAP=<N>"): tools/ab_gap.sh "2 20" g6 g10 ...
spec=${1:-"2 20"}; shift
for rep in 1 2; do
for t in default "$@"; do
lib=""; [ "$t" != default ] && lib=$PWD/tools/bin/libhxsht_$t.so
set -- $spec "$@"
env HX_LIBRARY=$lib NSIDE=4096 LMAX=6144 SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>/dev/null | sed "s|^|$t: |" | cut -c1-140
shift 2
done; done
