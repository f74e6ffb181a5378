#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_y; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
python -m pytest tests -m gpu -q -k "hash or thumb or switch" 2>&1 | grep -E "passed|failed|rror|FAILED" | tail -5
for i in 1 2; do for w in 1 0 8 16 64; do
  if [ $w = 0 ]; then unset PSD_HASH_WALK; else export PSD_HASH_WALK=$w; fi
  python bench.py --detector hash --no-secondary --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hash walk $w', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['value'])"
done; done 2>&1 | tee $O/ab_hash_walk.txt
