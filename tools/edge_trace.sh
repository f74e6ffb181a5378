#!/bin/bash
# per-kernel times of the edge term + HSV (one read) on 1024 x 1080p frames of each content kind: S (shots, no edges),
# T (objects: real Canny edges), U (uniform noise).  usage: tools/edge_trace.sh <outdir> [tag]
O=${1:-gpurun_out/edge_trace}; TAG=${2:-}; mkdir -p $O
R=$PWD; export PYTHONPATH=$R:$R/tools
cd /tmp; export TMPDIR=/tmp
for d in S T U; do
  rm -rf /tmp/etrace
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/etrace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 1 --detector edges --dist $d --frames 1024 > $R/$O/bench_edges_$d$TAG.json 2>/dev/null
  python $R/tools/kernel_stats_md.py /tmp/etrace/t_kernel_stats.csv "edges + HSV, dist $d, 1024 x 1080p, 5 steps $TAG" > $R/$O/kernel_trace_edges_$d$TAG.md 2>&1
  head -12 $R/$O/kernel_trace_edges_$d$TAG.md | cut -c1-150
  python -c "
import json; d=json.load(open('$R/$O/bench_edges_$d$TAG.json')); print('$d', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done
