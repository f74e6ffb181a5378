#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_ag}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
t() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 8 --warmup 2 $2 2>>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-26s %-44s' % ('$1', '$2'), d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $O/ab.txt; }
for e in X=0 PSD_EDGE_WS_MB=8192 PSD_EDGE_WS_MB=16384 X=0 PSD_EDGE_WS_MB=16384; do
  t $e "--detector edges --dist S --frames 2048"
  t $e "--detector edges --dist S --frames 4096"
done
