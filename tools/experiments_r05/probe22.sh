#!/bin/bash
# round 5, probe 22: large record copies on a stream of their own (PSD_D2H_STREAM, default 1) against the scoring stream (0): GPU tests, then
# whole-step rates of the workloads whose records carry histograms, alternating.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export PYTHONPATH=$R:$R/tools
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], (d.get('parity_sample') or '')[-30:])"; }
for rep in 1 2; do for k in 1 0; do
  export PSD_D2H_STREAM=$k
  python bench.py --detector all --no-cpu-baseline --no-secondary --steps 12 2>/dev/null | tail -1 | line "all d2h=$k"
  python bench.py --detector all --downscale auto --no-cpu-baseline --no-secondary --steps 30 2>/dev/null | tail -1 | line "downscale_all d2h=$k"
  python bench.py --detector hist --res 4k --frames 2048 --no-cpu-baseline --no-secondary --steps 8 2>/dev/null | tail -1 | line "hist4k d2h=$k"
  python bench.py --workload corpus --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | line "corpus d2h=$k"
  python bench.py --workload bbc --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | line "bbc d2h=$k"
done; done
