#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tools
for n in -1 0 1 -1; do python tools/feed_numa.py $n 2>/dev/null | tail -1; done
PSD_FEED_NUMA=0 python tools/feed_numa.py -1 2>/dev/null | tail -1
python bench.py --detector all --downscale auto --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('downscale all:', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
timeout 300 python -m pytest tests/test_gpu_feed_rows.py -q -m gpu 2>&1 | tail -2
