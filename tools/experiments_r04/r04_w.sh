#!/bin/bash
# round 4, step w: histogram replicas of the fused 16-wave passes (PSD_FUSED_AC = 4 / 8 / 16) on S / T / U / K, V mode (edge term) and all-detectors
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_w; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
A=$R/pyscenedetect_amd/csrc/build/abl
{ python tools/edge_ab.py 2048 STUK ac8; PSD_LIB_PATH=$A/libpsd_ac16.so python tools/edge_ab.py 2048 STUK ac16; PSD_LIB_PATH=$A/libpsd_ac4.so python tools/edge_ab.py 2048 STUK ac4; python tools/edge_ab.py 2048 STUK ac8; PSD_LIB_PATH=$A/libpsd_ac16.so python tools/edge_ab.py 2048 STUK ac16; } 2>&1 | grep -v amdgpu.ids | tee $O/edge_ab.txt
for d in S K; do echo "all four fused, dist $d"; bash tools/ab_libs.sh "--detector all --dist $d" ac16 ac4; done 2>&1 | tee $O/fused_ab.txt
