#!/bin/bash
# round 5, probe 19: how the records of a SMALL submission reach the host (PSD_SMALL_COPY 0 kernel stores into the pinned mirror, 1 strided
# 2-D copy, 2 pack + contiguous copy) x completion by polling or by the runtime's wait (PSD_SPIN_US 400 / 0): per-frame API times
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export PYTHONPATH=$R:$R/tools
for rep in 1 2; do for c in 0 1 2; do for k in 400 0; do
  echo -n "copy=$c spin=$k: "; PSD_SMALL_COPY=$c PSD_SPIN_US=$k timeout 120 python tools/experiments_r05/per_frame_breakdown.py 2>/dev/null | tail -1
done; done; done
