#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_ak}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
PSD_EDGE_SPECULATIVE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 --detector edges --dist T --frames 1024 > /dev/null 2>&1
python $R/tools/kernel_stats_md.py $O/trace/t_kernel_stats.csv "edges+HSV T 1024, host-driven loop, 3 steps" | cut -c1-170 | grep -E "hysteresis|sobel" | tee $O/trace_T_exact.md
rm -rf $O/trace
cd $R
timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 16 --warmup 8 --detector edges --dist T --frames 1024 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('edges+HSV T 1024 after 8 warm-up steps', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
