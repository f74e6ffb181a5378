#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_t}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
PSD_LIB_PATH=$R/pyscenedetect_amd/csrc/build/abl/libpsd_w6.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=600 > $O/pytest_w6.log 2>&1; echo "pytest(w6) rc=$?"; grep -E "passed|failed" $O/pytest_w6.log | tail -1
t() { env $4 PSD_LIB_PATH=$2 timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 $3 2>>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s %-14s %-40s' % ('$1', '$4', '$3'), d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $O/ab.txt; }
for v in default w6 default w6; do
  L=$R/pyscenedetect_amd/csrc/build/abl/libpsd_$v.so; [ $v = default ] && L=$R/pyscenedetect_amd/libpsd_hip.so
  t $v $L "" X=0
  t $v $L "--dist S --frames 2048" X=0
  t $v $L "--height 2160 --width 3840 --frames 1024" X=0
done
t default $R/pyscenedetect_amd/libpsd_hip.so "" PSD_SCORE_G=2
t default $R/pyscenedetect_amd/libpsd_hip.so "" PSD_HSV_BLOCKS=16384
t default $R/pyscenedetect_amd/libpsd_hip.so "" PSD_HSV_BLOCKS=65536
