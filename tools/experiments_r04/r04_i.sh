#!/bin/bash
out=gpurun_out/r04_i; mkdir -p $out
export PYTHONPATH=$PWD:$PWD/tools
timeout 1200 python -m pytest tests -x -q -m gpu --durations=6 > $out/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.txt; tail -14 $out/pytest_gpu.txt
for d in all; do
  timeout 300 python bench.py --detector $d --downscale auto --no-secondary --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_ds_$d.json 2>$out/bench_ds_$d.err
  python -c "
import json; d=json.load(open('$out/bench_ds_$d.json')); print('$d', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done
bash tools/edge_trace.sh $out _cleanup | grep -v "rocclr\|store_xor\|median\|^$\|^|---\|^| kernel\|dilate"
