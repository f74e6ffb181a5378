#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_ad}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
t() { PSD_LIB_PATH=$2 timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 $3 2>>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s %-52s' % ('$1', '$3'), d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $O/ab.txt; }
for v in default fswap default fswap; do
  L=$R/pyscenedetect_amd/csrc/build/abl/libpsd_$v.so; [ $v = default ] && L=$R/pyscenedetect_amd/libpsd_hip.so
  t $v $L "--detector all"
  t $v $L "--detector all --dist S --frames 2048"
done
