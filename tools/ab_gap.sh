for spec in "2 18" "0 16" "0 10" "2 12" "2 14"; do set -- $spec
for t in default nogap g8all; do
lib=""; [ "$t" != default ] && lib=$PWD/tools/bin/libhxsht_$t.so
env HX_LIBRARY=$lib NSIDE=4096 LMAX=6144 SPIN=$1 NCOMP=$2 python tools/leg_only.py 2>/dev/null | sed "s|^|$t: |" | cut -c1-140
done; done
