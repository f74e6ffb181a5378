#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_d}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^(FAILED|ERROR)" $O/pytest.log | head -20
t() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2 $2 2>>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-26s %-40s' % ('$1', '$2'), d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $O/ab.txt; }
for e in PSD_X=0 PSD_EDGE_VHIST_FUSED=0; do
  t $e "--detector edges --dist S --frames 2048"
  t $e "--detector edges --dist U --frames 1024"
done
( for n in 64 256 1024; do ET_N=$n ET_SMOOTH=1 timeout 200 python tools/edge_time.py 2>/dev/null | tail -1 | sed "s/^/shot-like /"; done; ET_N=256 timeout 200 python tools/edge_time.py 2>/dev/null | tail -1 | sed "s/^/uniform noise /" ) | tee $O/edge_time.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 1 --detector edges --dist S --frames 2048 > /dev/null 2>&1
python $R/tools/kernel_stats_md.py $O/trace/t_kernel_stats.csv "edges+HSV S 2048" 2>/dev/null | head -20 | cut -c1-200 | tee $O/trace_S.md
rm -rf $O/trace
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 1 --detector edges --dist U --frames 1024 > /dev/null 2>&1
python $R/tools/kernel_stats_md.py $O/trace/t_kernel_stats.csv "edges+HSV U 1024" 2>/dev/null | head -16 | cut -c1-200 | tee $O/trace_U.md
rm -rf $O/trace
