#!/bin/bash
# SQ counters of the edge kernels on frames with objects (--dist T) against the edge-free shots (--dist S): where the
# Sobel / NMS kernel's extra time on T comes from (instructions or waiting)
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_av}; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for d in S T; do
  timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
     -d $O/pmc_$d/sq1 -o pmc --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 --detector edges --dist $d --frames 1024 > $O/pmc_$d.log 2>&1
  echo "dist $d rc=$?"
  for k in sobel_nms hysteresis_frame dilate_xor score_frames; do echo "== $d $k"; python $R/tools/pmc_summary.py $O/pmc_$d $k; done
done > $O/pmc_edges_S_vs_T.txt 2>&1
cat $O/pmc_edges_S_vs_T.txt | cut -c1-120
rm -rf $O/pmc_S $O/pmc_T
