# same-device A/B of k_synth_duo builds (tools/build_variant.sh hx_synth_duo.hip <tag> "<flags>"): tools/ab_synth.sh "2:10,0:10" tag1 tag2 ...
cases=${1:-"2:10,0:10"}; shift
for rep in 1 2; do
for t in default "$@"; do
lib=""; [ "$t" != default ] && lib=$PWD/tools/bin/libhxsht_$t.so
env HX_LIBRARY=$lib CASES=$cases python tools/time_synth_duo.py 2>/dev/null | sed "s|^|$t: |" | cut -c1-150
done; done
