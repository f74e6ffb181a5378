#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_j}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 600 python -c "
import cProfile, pstats, sys, io
sys.argv=['bench.py','--workload','bbc','--steps','4','--warmup','1','--no-cpu-baseline']
import bench
pr=cProfile.Profile(); pr.enable()
try:
    bench.main()
finally:
    pr.disable(); s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('cumulative').print_stats(45); open('$O/prof_bbc.txt','w').write(s.getvalue())
" > $O/bbc.json 2> $O/bbc.err; tail -c 600 $O/bbc.json; grep -v "^$" $O/prof_bbc.txt | head -70 | cut -c1-170
for d in U S; do for n in 2048 4096; do timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 8 --warmup 2 --dist $d --frames $n 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$d $n', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done; done | tee $O/size_dist.txt
