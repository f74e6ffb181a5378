#!/bin/bash
# Build ablated variants of the scoring kernel (timing experiments; wrong results by design).
set -e
cd "$(dirname "$0")/../pyscenedetect_amd/csrc"
mkdir -p build/abl
MACRO=PSD_ABLATE; TAG=abl
# tools/ablate.sh -f NAME "-DPSD_FUSED_LC=8 -DPSD_FUSED_F=2"   builds libpsd_NAME.so with extra defines
if [ "$1" = "-f" ]; then
  mkdir -p build/abl
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -I. $3 -c psd_score_kernels.hip -o build/abl/score_$2.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libpsd_$2.so build/abl/score_$2.o build/psd_edge_kernels.hip.o build/psd_hash_kernels.hip.o build/psd_resize_kernels.hip.o build/psd_engine.cpp.o build/psd_feed.cpp.o build/psd_epilogue.cpp.o build/psd_comm.cpp.o -ldl
  echo built $2; exit 0
fi
for a in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -I. -D$MACRO=$a -c psd_score_kernels.hip -o build/abl/score_$TAG$a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libpsd_$TAG$a.so build/abl/score_$TAG$a.o build/psd_edge_kernels.hip.o build/psd_hash_kernels.hip.o build/psd_resize_kernels.hip.o build/psd_engine.cpp.o build/psd_feed.cpp.o build/psd_epilogue.cpp.o build/psd_comm.cpp.o -ldl
  echo built $TAG$a
done
