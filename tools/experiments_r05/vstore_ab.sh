#!/bin/bash
# NEXT ROUND, first GPU call: why does the V-plane store of the V-mode front end cost 0.17 ms per 1024 shot-like frames and 0.04 ms
# on noise (profiles/r04_ag_*)?  Builds the three store variants of psd_score_kernels.hip (PSD_VSTORE_MODE: 1 = non-temporal,
# 2 = the bytes scrambled with their position -- WRONG edge results by design, the histogram stays right so nothing hangs --,
# 3 = sc0 sc1) and times the edge term on S / T / U against the default, alternating, one process per library.
# Build here (no GPU needed):  bash tools/next_vstore_ab.sh build      On the GPU box:  bash tools/next_vstore_ab.sh
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export PYTHONPATH=$R:$R/tools
if [ "$1" = build ]; then
  for m in 1 2 3 4 5; do bash tools/ablate.sh -f vs$m "-DPSD_VSTORE_MODE=$m"; done; exit 0
fi
O=$R/gpurun_out/next_vstore; mkdir -p $O; A=$R/pyscenedetect_amd/csrc/build/abl
for rep in 1 2; do
  timeout 120 python tools/edge_ab.py 2048 STU default
  for m in 1 2 3 4 5; do [ -f $A/libpsd_vs$m.so ] && PSD_LIB_PATH=$A/libpsd_vs$m.so timeout 120 python tools/edge_ab.py 2048 STU vs$m; done
done 2>&1 | grep -v amdgpu.ids | tee $O/vstore_ab.txt
