#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_i}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
( timeout 600 python tools/walk_sweep.py
  for b in 8192 16384 65536; do PSD_HSV_BLOCKS=$b timeout 300 python tools/walk_sweep.py 1080x1920x2048 1080x1920x4096 360x640x36864; done ) 2>&1 | grep -v Warning | tee $O/walk_sweep.txt
