#!/bin/sh
# One translation unit rebuilt with extra flags and linked with the current objects of the library, for A/B runs on one device
# (HX_LIBRARY=tools/bin/libhxsht_<tag>.so):   tools/build_variant.sh hx_analysis.hip latebar0 "-DHX_PIPE_LATEBAR=0"
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
src=$1; tag=$2; flags=$3
mkdir -p $ROOT/tools/bin
cd $ROOT/heracles_amd/csrc
make -s -j4 >/dev/null
base=$(basename $src .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c $src -o /tmp/${base}_$tag.o
objs=$(ls *.o | grep -v "^$base.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/bin/libhxsht_$tag.so $objs /tmp/${base}_$tag.o
echo built tools/bin/libhxsht_$tag.so
