#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_p}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^(FAILED|ERROR)" $O/pytest.log | head -20; grep -E "^E " $O/pytest.log | head -20
t() { PSD_LIB_PATH=$2 timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 $3 2>>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s %-52s' % ('$1', '$3'), d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $O/ab.txt; }
for v in old default old default; do
  L=$R/pyscenedetect_amd/csrc/build/abl/libpsd_$v.so; [ $v = default ] && L=$R/pyscenedetect_amd/libpsd_hip.so
  t $v $L "--detector hist --height 2160 --width 3840 --frames 1024"
  t $v $L "--detector hash"
  t $v $L "--downscale auto"
  ( export PSD_LIB_PATH=$L; ET_N=1024 ET_SMOOTH=1 timeout 200 python tools/edge_time.py 2>/dev/null | tail -1 | sed "s/^/$v shot-like /" ) | tee -a $O/ab.txt
done
