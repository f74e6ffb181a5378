#!/bin/bash
# BBC stand-in flow (640x360 -> 256x144): destination rows per workgroup x whole rounds per launch
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_v; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
for i in 1 2; do for rows in 0 2; do for rounds in 4 6 8 10; do
  if [ $rows = 0 ]; then unset PSD_RESIZE_ROWS; else export PSD_RESIZE_ROWS=$rows; fi
  PSD_RESIZE_ROUNDS=$rounds python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 10 --workload bbc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rows $rows rounds $rounds', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['value'])"
done; done; done 2>&1 | tee $O/ab_bbc.txt
unset PSD_RESIZE_ROWS
for i in 1 2; do for rounds in 4 6 8; do
  PSD_RESIZE_ROUNDS=$rounds python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 10 --workload corpus 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('corpus rounds $rounds', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['value'])"
done; done 2>&1 | tee $O/ab_corpus.txt
