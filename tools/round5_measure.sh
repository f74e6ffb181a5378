#!/bin/bash
# Round-5 measurement matrix on one MI355X: the round-4 matrix (tools/round4b_measure.sh: GPU tests, smoke, default bench x 3,
# flows, 1-rank launcher run, the secondary kernels on their own, kernel traces, PMC passes for HBM traffic, feed rates) under
# the tag given, then what round 5 added: the flows' GPU timelines (rocprofv3 kernel + memory-copy trace -> busy / idle time
# per pass) and the per-frame API's time by step.   usage: tools/round5_measure.sh <tag>
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r05_m}; O=$R/gpurun_out/$T; mkdir -p $O
bash $R/tools/round4b_measure.sh $T
cd $R; export PYTHONPATH=$R:$R/tools
timeout 120 python tools/experiments_r05/per_frame_breakdown.py 2>/dev/null | tail -1 > $O/per_frame_api_breakdown.txt; cat $O/per_frame_api_breakdown.txt
cd /tmp; export TMPDIR=/tmp
for w in bbc corpus; do
  timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d $O/tl_$w -o t --output-format csv -- python $R/bench.py --workload $w --bbc-frames 2000 --corpus-frames 512 --steps 3 --warmup 2 --no-cpu-baseline > $O/flow_small_$w.json 2> /dev/null
  python $R/tools/experiments_r05/timeline.py $O/tl_$w 14 > $O/timeline_$w.txt 2>&1; tail -1 $O/timeline_$w.txt
  rm -rf $O/tl_$w
done
ls $O | wc -l
