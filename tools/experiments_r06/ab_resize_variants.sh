#!/bin/bash
# Round 6, second session: variants of resize_walk_kernel on the flows it decides (BBC stand-in 640x360, corpus 1080p + 4K, 1080p device batch).
#   swap   = two steps per trip with swapped H,S,V register sets (-DPSD_RS_SWAP=1)
#   hue    = the hue's case distinction as selects instead of divergent branches (-DPSD_RS_HUE_SELECT=1)
#   wg512  = 8-wave workgroups (-DPSD_RS_WG=512)
# plus PSD_RESIZE_ROWS on the in-tree build.  usage: tools/experiments_r06/ab_resize_variants.sh <tag>
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r06_q}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
V="default swap hue swaphue wg512"
for v in swap hue swaphue wg512; do
  echo "== parity $v"; PSD_LIB_PATH=$R/pyscenedetect_amd/csrc/build/abl/libpsd_$v.so timeout 600 python -m pytest tests/test_gpu_corpus_default.py tests/test_gpu_parity.py -m gpu -x -q -k "downscale or corpus or default or packed or clips or resize" 2>&1 | tail -2
done > $O/parity.txt 2>&1
cat $O/parity.txt
{ echo "## bbc"; bash tools/ab_libs_long.sh "--workload bbc" 2 $V
  echo "## corpus"; bash tools/ab_libs_long.sh "--workload corpus" 2 $V
  echo "## 1080p content behind auto downscale"; bash tools/ab_libs_long.sh "--frames 4096 --downscale auto" 2 $V
  echo "## 1080p all four behind auto downscale"; bash tools/ab_libs_long.sh "--frames 4096 --downscale auto --detector all" 2 $V
  for rows in 2 3 4 6; do echo "## bbc PSD_RESIZE_ROWS=$rows"; PSD_RESIZE_ROWS=$rows bash tools/ab_libs_long.sh "--workload bbc" 2 default; done
  for rows in 1 2 3; do echo "## 1080p PSD_RESIZE_ROWS=$rows"; PSD_RESIZE_ROWS=$rows bash tools/ab_libs_long.sh "--frames 4096 --downscale auto" 2 default; done
} 2>&1 | tee $O/ab.txt
