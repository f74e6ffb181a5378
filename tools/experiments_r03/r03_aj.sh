#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_aj}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^(FAILED|ERROR)" $O/pytest.log | head -20; grep -E "^E " $O/pytest.log | head -20
for d in S T U; do timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2 --detector edges --dist $d --frames 1024 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('edges+HSV $d 1024', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('parity_sample'))"; done | tee $O/ab.txt
( for n in 256 1024; do ET_N=$n ET_SMOOTH=1 timeout 200 python tools/edge_time.py 2>/dev/null | tail -1 | sed "s/^/shot-like /"; done; ET_N=256 timeout 200 python tools/edge_time.py 2>/dev/null | tail -1 | sed "s/^/uniform noise /" ) | tee -a $O/ab.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 --detector edges --dist T --frames 1024 > /dev/null 2>&1
python $R/tools/kernel_stats_md.py $O/trace/t_kernel_stats.csv "edges+HSV T 1024, 4 steps" | cut -c1-170 | tee $O/trace_T.md
rm -rf $O/trace
