#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_ai}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 --detector edges --dist T --frames 1024 > $O/bench_T.json 2>/dev/null
python $R/tools/kernel_stats_md.py $O/trace/t_kernel_stats.csv "edges+HSV T 1024, 4 steps" | cut -c1-170 | tee $O/trace_T.md
rm -rf $O/trace
cd $R; PSD_EDGE_SPECULATIVE=0 timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 1 --detector edges --dist T --frames 1024 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('host-driven loop: T 1024', d['value'], d['roofline']['avg_launch_ms'])"
