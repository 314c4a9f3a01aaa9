#!/bin/bash
# Diagnostic builds of libhxsht.so with phases of the pipelined Legendre kernel removed (results wrong by design):
#   tools/bin/libhxsht_abl<N>.so, N = bit mask HX_PIPE_ABL (1 no MFMA, 2 no recursion, 4 no flush).
# Select one with HX_LIBRARY=tools/bin/libhxsht_abl<N>.so (tools/leg_only.py).
set -e
cd "$(dirname "$0")/.."
make -C heracles_amd/csrc -j8 >/dev/null
mkdir -p tools/bin
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DHX_DIAG -DHX_PIPE_ABL=$n -c heracles_amd/csrc/hx_analysis.hip -o tools/bin/hx_analysis_abl$n.o
  objs=$(ls heracles_amd/csrc/*.o | grep -v hx_analysis.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libhxsht_abl$n.so $objs tools/bin/hx_analysis_abl$n.o
done
