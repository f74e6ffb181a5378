#!/bin/bash
# Round 6, second session: how many whole rounds of resident workgroups a launch of resize_walk_kernel is cut into
# (PSD_RESIZE_ROUNDS: 0 = the rule until now, 12 workgroups per CU rounded up to whole chunks per tile; unset = the new default)
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r06_r}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
timeout 900 python -m pytest tests/test_gpu_corpus_default.py tests/test_gpu_parity.py tests/test_gpu_headline_geometry.py -m gpu -x -q 2>&1 | tail -2 > $O/parity.txt; cat $O/parity.txt
run() { for i in 1 2; do for r in 0 1 2 3 4 6 8 default; do
    if [ $r = default ]; then unset PSD_RESIZE_ROUNDS; else export PSD_RESIZE_ROUNDS=$r; fi
    python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 10 $1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rounds %-8s' % '$r', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['value'])"
  done; done; unset PSD_RESIZE_ROUNDS; }
{ echo "## bbc"; run "--workload bbc"
  echo "## corpus"; run "--workload corpus"
  echo "## 1080p content behind auto downscale"; run "--frames 4096 --downscale auto"
  echo "## 1080p all four behind auto downscale"; run "--frames 4096 --downscale auto --detector all"
  echo "## 4K content behind auto downscale"; run "--frames 1024 --height 2160 --width 3840 --downscale auto"
} 2>&1 | tee $O/ab.txt
