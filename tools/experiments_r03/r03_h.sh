#!/bin/bash
# round 3, call h: GPU tests (incl. the environment-switch probes), the tail schedule of the time walks (A/B of library
# builds), the Sobel kernel after the unconditional tile loads, SQ counters of the edge kernels.
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_h}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^(FAILED|ERROR)" $O/pytest.log | head -20
PSD_LIB_PATH=$R/pyscenedetect_amd/csrc/build/abl/libpsd_t4.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_flows.py -m gpu -q --timeout=600 > $O/pytest_t4.log 2>&1; echo "pytest(t4) rc=$?"; grep -E "passed|failed" $O/pytest_t4.log | tail -2; grep -E "^(FAILED|ERROR)" $O/pytest_t4.log | head
t() { PSD_LIB_PATH=$2 timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 8 --warmup 2 $3 2>>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s %-52s' % ('$1', '$3'), d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $O/ab.txt; }
for v in default t2 t4 t8 t4r2 default; do
  L=$R/pyscenedetect_amd/csrc/build/abl/libpsd_$v.so; [ $v = default ] && L=$R/pyscenedetect_amd/libpsd_hip.so
  t $v $L ""
  t $v $L "--dist S --frames 2048"
  t $v $L "--detector all"
  t $v $L "--height 360 --width 640 --frames 36864 --dist S"
done
( for n in 64 256 1024; do ET_N=$n ET_SMOOTH=1 timeout 200 python tools/edge_time.py 2>/dev/null | tail -1 | sed "s/^/shot-like /"; done; ET_N=256 timeout 200 python tools/edge_time.py 2>/dev/null | tail -1 | sed "s/^/uniform noise /" ) | tee $O/edge_time.txt
bash tools/pmc_edges.sh $T > /dev/null 2>&1; cp $O/pmc_edges.txt $O/pmc_edges_sq.txt; rm -rf $O/pmc_edges; cut -c1-150 $O/pmc_edges_sq.txt | head -70
