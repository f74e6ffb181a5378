#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_v}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
export PSD_LIB_PATH=$R/pyscenedetect_amd/csrc/build/abl/libpsd_tl.so
for n in 1024 4096; do timeout 300 python tools/wg_timeline.py $n hsv 2>&1 | grep -v Warning | tee -a $O/timeline.txt; done
timeout 300 python tools/wg_timeline.py 4096 all 2>&1 | grep -v Warning | tee -a $O/timeline.txt
