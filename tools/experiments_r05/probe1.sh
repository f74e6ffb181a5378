#!/bin/bash
# round 5, first probe: differential fuzz, the per-frame API's time by step, GPU timelines of the two small flows
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
timeout 200 python tools/fuzz_gpu.py --seconds 100 --seed 1 > $O/fuzz_seed1.json 2> $O/fuzz_seed1.err; echo "fuzz rc=$?"; cut -c1-1500 $O/fuzz_seed1.json; tail -3 $O/fuzz_seed1.err
timeout 120 python tools/experiments_r05/per_frame_breakdown.py > $O/per_frame_breakdown.txt 2>&1; tail -2 $O/per_frame_breakdown.txt
cd /tmp; export TMPDIR=/tmp
for w in bbc corpus; do
  timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d $O/tl_$w -o t --output-format csv -- python $R/bench.py --workload $w --bbc-frames 2000 --corpus-frames 512 --steps 3 --warmup 2 --no-cpu-baseline > $O/flow_$w.json 2> $O/flow_$w.err
  python $R/tools/experiments_r05/timeline.py $O/tl_$w 14 > $O/timeline_$w.txt 2>&1; tail -60 $O/timeline_$w.txt | cut -c1-150
  python -c "import json; d=json.load(open('$O/flow_$w.json')); print('$w', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  rm -rf $O/tl_$w
done
