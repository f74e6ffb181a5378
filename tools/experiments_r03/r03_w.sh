#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_w}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
for v in tl0 tl; do echo "== $v"; PSD_LIB_PATH=$R/pyscenedetect_amd/csrc/build/abl/libpsd_$v.so timeout 300 python tools/wg_timeline.py 4096 hsv 2>&1 | grep -v Warning | grep -E "kernel|tail|resident" | sed -n '1p;18,22p'; done | tee $O/timeline.txt
t() { PSD_LIB_PATH=$2 timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 $3 2>>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s %-52s' % ('$1', '$3'), d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $O/ab.txt; }
for v in nofill default nofill default; do
  L=$R/pyscenedetect_amd/csrc/build/abl/libpsd_$v.so; [ $v = default ] && L=$R/pyscenedetect_amd/libpsd_hip.so
  t $v $L ""
  t $v $L "--dist S --frames 2048"
  t $v $L "--frames 1024"
  t $v $L "--height 2160 --width 3840 --frames 1024"
  t $v $L "--height 360 --width 640 --frames 36864 --dist S"
done
