#!/bin/bash
# A/B of library builds (tools/ablate.sh -f NAME ...) on the fused all-detectors pass and the edges + HSV workload.
# usage: tools/r03_ab.sh <tag> name1 name2 ...
R=${GRAFT_REPO_ROOT:-$PWD}; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
t() { PSD_LIB_PATH=$2 timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2 $3 2>>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-12s %-40s' % ('$1', '$3'), d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $O/ab.txt; }
for v in default "$@" default; do
  L=$R/pyscenedetect_amd/csrc/build/abl/libpsd_$v.so; [ $v = default ] && L=$R/pyscenedetect_amd/libpsd_hip.so
  t $v $L "--detector all"
  t $v $L "--detector all --dist S --frames 2048"
  t $v $L "--detector edges --dist S --frames 2048"
done
