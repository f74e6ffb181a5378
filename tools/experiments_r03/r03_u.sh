#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_u}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
t() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 $2 2>>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-24s %-48s' % ('$1', '$2'), d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $O/ab.txt; }
for b in X=0 PSD_HSV_BLOCKS=24576 PSD_HSV_BLOCKS=16384 PSD_HSV_BLOCKS=12288 X=0; do
  t $b ""
  t $b "--dist S --frames 2048"
  t $b "--height 2160 --width 3840 --frames 1024"
  t $b "--height 360 --width 640 --frames 36864 --dist S"
  t $b "--frames 1024"
done
