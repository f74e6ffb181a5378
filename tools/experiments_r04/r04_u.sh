#!/bin/bash
# round 4, step u: the V-mode front end on constant frames (K: every lane of a wave counts the same histogram bin)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_u; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
cd /tmp; export TMPDIR=/tmp
for d in K U; do
  rm -rf /tmp/etrace
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/etrace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 1 --detector edges --dist $d --frames 1024 > $O/bench_edges_$d.json 2>/dev/null
  python $R/tools/kernel_stats_md.py /tmp/etrace/t_kernel_stats.csv "edges + HSV, dist $d, 1024 x 1080p, 5 steps" > $O/kernel_trace_edges_$d.md 2>&1
  head -9 $O/kernel_trace_edges_$d.md | cut -c1-150
done
