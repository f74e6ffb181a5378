#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_c}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 -k "edge or Edge or flows or fullsize or parity" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
t() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2 $2 2>>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-26s %-40s' % ('$1', '$2'), d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $O/ab.txt; }
for e in PSD_X=0 PSD_EDGE_VHIST_FUSED=1 PSD_X=0 PSD_EDGE_VHIST_FUSED=1; do
  t $e "--detector edges --dist S --frames 2048"
  t $e "--detector edges --dist U --frames 1024"
done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 1 --detector edges --dist S --frames 2048 > /dev/null 2>&1
python $R/tools/kernel_stats_md.py $O/trace/t_kernel_stats.csv "edges+HSV S 2048" 2>/dev/null | head -24 | cut -c1-200
rm -rf $O/trace
