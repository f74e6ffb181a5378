#!/bin/bash
# Build a variant of libpsd_hip.so for A/B runs on ONE box (PSD_LIB_PATH selects it; tools/ab_libs.sh times it).
#   tools/build_variant.sh NAME FILE.hip "-DMACRO=1 ..." [GIT_REV]
# compiles FILE.hip (from GIT_REV of this repository if given, else the working tree) with the extra defines, links it with the
# other objects of the current build into pyscenedetect_amd/csrc/build/abl/libpsd_NAME.so.  Experiments only.
set -e
cd "$(dirname "$0")/../pyscenedetect_amd/csrc"
NAME=$1; FILE=$2; DEFS=$3; REV=$4
mkdir -p build/abl
SRC=$FILE
if [ -n "$REV" ]; then
  SRC=build/abl/${NAME}_$FILE
  git show "$REV:pyscenedetect_amd/csrc/$FILE" > "$SRC"
fi
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -I. $DEFS -c "$SRC" -o build/abl/${NAME}_$FILE.o
OBJS=""
for f in psd_score_kernels.hip psd_edge_kernels.hip psd_hash_kernels.hip psd_resize_kernels.hip psd_engine.cpp psd_feed.cpp psd_epilogue.cpp psd_comm.cpp; do
  if [ "$f" = "$FILE" ]; then OBJS="$OBJS build/abl/${NAME}_$FILE.o"; else OBJS="$OBJS build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libpsd_$NAME.so $OBJS -ldl
echo built build/abl/libpsd_$NAME.so
