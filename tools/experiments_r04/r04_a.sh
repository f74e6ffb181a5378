#!/bin/bash
# round 4, GPU call a: new parity tests at BASELINE's launch geometry, the batched row feeder, feed rates
out=gpurun_out/r04_a; mkdir -p $out
export PYTHONPATH=$PWD:$PWD/tools
timeout 900 python -m pytest tests/test_gpu_headline_geometry.py tests/test_gpu_feed_rows.py -x -q -m gpu --durations=12 > $out/pytest_new.txt 2>&1
echo "pytest rc=$?" >> $out/pytest_new.txt
tail -25 $out/pytest_new.txt
for t in 8 16; do
  PSD_FEED_THREADS=$t timeout 600 python tools/feed_bench.py > $out/feed_threads_$t.json 2> $out/feed_threads_$t.err
  echo "feed $t rc=$?"; cat $out/feed_threads_$t.json
done
nproc; lscpu | grep -i "model name\|socket\|numa" | head -5
