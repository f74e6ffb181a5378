#!/bin/bash
# round 5, probe 14: per-kernel times of the edge term with the V plane keyed (PSD_EDGE_VKEY=1, default) and plain (0), 1024 x 1080p
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05o; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
cd /tmp; export TMPDIR=/tmp
for d in S T; do for k in 1 0 1 0; do
  rm -rf /tmp/etrace
  PSD_EDGE_VKEY=$k timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/etrace -o t --output-format csv -- python $R/tools/edge_ab.py 1024 $d vkey$k > $O/ab_${k}_$d.log 2>&1
  grep "records crc" $O/ab_${k}_$d.log | cut -c1-150
  python - <<PY
import csv
for r in csv.DictReader(open('/tmp/etrace/t_kernel_stats.csv')):
    n = r['Name']
    if any(x in n for x in ('score_frames_dma','sobel','hysteresis','dilate','median')): print('vkey$k $d', n[:44], 'avg us', round(float(r['AverageNs'])/1e3,1), 'calls', r['Calls'])
PY
done; done 2>&1 | tee $O/vkey_kernels.txt
