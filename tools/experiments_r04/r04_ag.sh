#!/bin/bash
# round 4, step ag: what the V-mode front end of the edge term pays on shot-like content: ablations (wrong results by design)
# 16 = no histogram increments, 32 = no V-plane store, 48 = neither, 2 = no per-frame barrier / flush
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_ag; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
A=$R/pyscenedetect_amd/csrc/build/abl
cd /tmp; export TMPDIR=/tmp
for v in default abl16 abl32 abl48 abl2; do for d in S U; do
  L=$A/libpsd_$v.so; [ $v = default ] && L=$R/pyscenedetect_amd/libpsd_hip.so
  rm -rf /tmp/etrace
  PSD_LIB_PATH=$L timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/etrace -o t --output-format csv -- python $R/tools/edge_ab.py 1024 $d $v > $O/ab_${v}_$d.log 2>&1; tail -2 $O/ab_${v}_$d.log | cut -c1-200; ls /tmp/etrace | head -3
  python - <<PY
import csv
for r in csv.DictReader(open('/tmp/etrace/t_kernel_stats.csv')):
    if 'score_frames_dma' in r['Name']: print('$v $d V-mode kernel avg us', round(float(r['AverageNs'])/1e3,1), 'calls', r['Calls'])
PY
done; done 2>&1 | tee $O/vmode_ablations.txt
