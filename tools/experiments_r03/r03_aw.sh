#!/bin/bash
# Sobel / NMS kernel with workgroups that walk the frames (the launch was paced by the dispatcher: two million one-microsecond waves)
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_aw}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout=300 -k "edge or Edge or edges" > $O/pytest_edges.log 2>&1; echo "edge tests rc=$?"; tail -2 $O/pytest_edges.log; grep -E "^E " $O/pytest_edges.log | head
for w in default 0; do
  for d in S T; do
    if [ $w = 0 ]; then export PSD_EDGE_SOBEL_WALKERS=0; else unset PSD_EDGE_SOBEL_WALKERS; fi
    timeout 200 python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2 --detector edges --dist $d --frames 1024 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('walkers=$w edges+HSV $d 1024', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
  done
done | tee $O/ab.txt
unset PSD_EDGE_SOBEL_WALKERS
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 --detector edges --dist T --frames 1024 > /dev/null 2>&1
python $R/tools/kernel_stats_md.py $O/trace/t_kernel_stats.csv "edges+HSV T 1024, walking Sobel kernel" | cut -c1-170 | head -9 | tee $O/trace_T.md
rm -rf $O/trace
