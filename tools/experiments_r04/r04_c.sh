#!/bin/bash
out=gpurun_out/r04_c; mkdir -p $out
export PYTHONPATH=$PWD:$PWD/tools
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_feed_rows.py tests/test_gpu_parity.py -x -q -m gpu -k "downscale or tap_rows or golden" > $out/pytest.txt 2>&1
echo "pytest rc=$?" >> $out/pytest.txt; tail -8 $out/pytest.txt
for d in content all hist; do
  timeout 300 python bench.py --detector $d --downscale auto --no-secondary --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_ds_$d.json 2>$out/bench_ds_$d.err
  python -c "
import json; d=json.load(open('$out/bench_ds_$d.json')); print('$d', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done
