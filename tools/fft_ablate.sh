#!/bin/bash
# builds libhxsht variants with phases of k_ring_subdft removed (HX_FFT_ABL bits: 1 no sincos,
# 2 no forward FFT, 4 no filter read, 8 no inverse FFT, 16 no pixel loads) and times the ring
# Fourier stage of one 8-component map2alm at nside 4096.  Results are wrong by design.
set -e
cd "$(dirname "$0")/../heracles_amd/csrc"
OBJS=$(ls *.o | grep -v "^hx_sht.o")
mkdir -p ../../tools/bin
for a in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DHX_FFT_ABL=$a -c hx_sht.hip -o /tmp/hx_sht_abl$a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libhxsht_fa$a.so $OBJS /tmp/hx_sht_abl$a.o
done
