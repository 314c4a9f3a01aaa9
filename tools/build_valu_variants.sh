#!/bin/sh
# Variants of the vector-unit Legendre kernel as separate libraries for A/B runs on one device (HX_LIBRARY=tools/bin/libhxsht_<tag>.so):
#   tools/build_valu_variants.sh tag "-DHX_VALU_WAVES=2" [tag2 "flags2" ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/tools/bin
cd $ROOT/heracles_amd/csrc
make -s -j4 >/dev/null
while [ $# -ge 2 ]; do
    tag=$1; flags=$2; shift; shift
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c hx_legendre_valu.hip -o /tmp/hx_legendre_valu_$tag.o
    objs=$(ls *.o | grep -v hx_legendre_valu.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/bin/libhxsht_$tag.so $objs /tmp/hx_legendre_valu_$tag.o
    echo built tools/bin/libhxsht_$tag.so
done
