#!/bin/bash
out=gpurun_out/r04_j; mkdir -p $out
R=$PWD; export PYTHONPATH=$R:$R/tools
cd /tmp; export TMPDIR=/tmp
for d in content all hist; do
  rm -rf /tmp/tr
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr -o t --output-format csv -- python $R/bench.py --detector $d --downscale auto --no-secondary --no-cpu-baseline --steps 20 --warmup 5 > $R/$out/bench_ds_$d.json 2>/dev/null
  python $R/tools/kernel_stats_md.py /tmp/tr/t_kernel_stats.csv "downscale auto, detector $d" > $R/$out/trace_ds_$d.md 2>&1
  grep -E "resize_walk|hist_reduce" $R/$out/trace_ds_$d.md | cut -c1-90,130-190
  python -c "
import json; d=json.load(open('$R/$out/bench_ds_$d.json')); print('$d', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done
