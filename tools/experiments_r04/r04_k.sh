#!/bin/bash
out=gpurun_out/r04_k; mkdir -p $out
export PYTHONPATH=$PWD:$PWD/tools
for t in 512 128 512 128; do
  PSD_HIST_REDUCE_THREADS=$t python bench.py --detector all --downscale auto --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('reduce threads $t:', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done
python bench.py --downscale auto --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('content:', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
for t in 16 32; do PSD_FEED_THREADS=$t python tools/feed_tune.py > $out/feed_tune_$t.json 2>$out/feed_tune_$t.err; cat $out/feed_tune_$t.json; echo; done
