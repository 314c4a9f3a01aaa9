#!/bin/sh
# One translation unit rebuilt with extra flags and linked with the current objects of the library, for A/B runs on one device
# (HX_LIBRARY=tools/bin/libhxsht_<tag>.so):   tools/build_variant.sh hx_analysis.hip latebar0 "-DHX_PIPE_LATEBAR=0"
#   tools/build_variant.sh --switches hx_analysis.hip abl2 "-DHX_DUO_ABL=2"
# builds from the sources WITH the diagnostic build switches of rounds 2-5, which left heracles_amd/csrc in round 6: the csrc of the base
# commit named in tools/patches/r05_switches.patch + that patch, in a scratch directory (the whole library is rebuilt there).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = "--switches" ]; then
  shift; src=$1; tag=$2; flags=$3
  base=$(sed -n 's/^# base commit: \([0-9a-f]*\).*/\1/p' $ROOT/tools/patches/r05_switches.patch | head -1)
  work=/tmp/hx_switch_src_$tag; rm -rf $work; mkdir -p $work $ROOT/tools/bin
  (cd $ROOT && git archive $base heracles_amd/csrc include) | tar -x -C $work
  grep -v '^#' $ROOT/tools/patches/r05_switches.patch | (cd $work && patch -s -p1)
  cd $work/heracles_amd/csrc
  for f in *.hip; do
    extra=""; [ "$f" = "$src" ] && extra="$flags"
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $extra -c $f -o ${f%.hip}.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/bin/libhxsht_$tag.so *.o
  echo built tools/bin/libhxsht_$tag.so from $base + r05_switches.patch
  exit 0
fi
src=$1; tag=$2; flags=$3
mkdir -p $ROOT/tools/bin
cd $ROOT/heracles_amd/csrc
make -s -j4 >/dev/null
base=$(basename $src .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c $src -o /tmp/${base}_$tag.o
objs=$(ls *.o | grep -v "^$base.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/bin/libhxsht_$tag.so $objs /tmp/${base}_$tag.o
echo built tools/bin/libhxsht_$tag.so
