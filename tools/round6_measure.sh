#!/bin/bash
# Round-6 measurement matrix on one MI355X: the round-4/5 matrix (tools/round4b_measure.sh: GPU tests, smoke, default bench x 3,
# flows -- since round 6 behind the reference's default downscale --, 1-rank launcher run, the secondary kernels on their own,
# kernel traces, PMC passes for HBM traffic, feed rates) under the tag given, then what round 6 added:
#   * the two flows at full resolution as well (--flow-pipeline full: what rounds 2-5 timed),
#   * the fused downscale kernel's HBM traffic (PMC: FETCH_SIZE / WRITE_SIZE, separate passes) for ContentDetector and for all four
#     detectors, and for the packed BBC flow (the SEG instance at 640x360),
#   * a kernel trace of the BBC flow (its dominant kernel is resize_walk_kernel<...,SEG> now),
#   * the default pipeline with the edge term, HashDetector with the device epilogue.
# usage: tools/round6_measure.sh <tag>
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r06_m}; O=$R/gpurun_out/$T; mkdir -p $O
bash $R/tools/round4b_measure.sh $T
cd $R; export PYTHONPATH=$R:$R/tools
for w in corpus bbc; do timeout 600 python bench.py --workload $w --flow-pipeline full --steps 6 --warmup 3 2>/dev/null | tail -1 > $O/bench_${w}_full_resolution.json; done
timeout 300 python bench.py --detector hash --no-secondary 2>/dev/null | tail -1 > $O/bench_hash.json
for d in S T U; do timeout 300 python bench.py --downscale auto --detector edges --dist $d --frames 4096 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_downscale_auto_edges_$d.json; done
for f in corpus_full_resolution bbc_full_resolution hash downscale_auto_edges_S downscale_auto_edges_T downscale_auto_edges_U; do python -c "import json; d=json.load(open('$O/bench_$f.json')); print('$f', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done
cd /tmp; export TMPDIR=/tmp
for what in downscale downscale_all bbc; do
  case $what in downscale) BA="--frames 4096 --steps 2 --warmup 1 --downscale auto";; downscale_all) BA="--frames 4096 --steps 2 --warmup 1 --downscale auto --detector all";;
    bbc) BA="--workload bbc --steps 2 --warmup 1";; esac
  P=$O/pmc_$what; mkdir -p $P
  for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --kernel-trace --pmc $c -d $P/$c -o pmc --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary $BA > $P/$c.log 2>&1; done
  python $R/tools/pmc_by_kernel.py $P resize_walk hist_reduce > $O/pmc_${what}_traffic.txt; cut -c1-250 $O/pmc_${what}_traffic.txt
  rm -rf $P
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_bbc -o t --output-format csv -- python $R/bench.py --workload bbc --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_bbc_under_rocprof.json 2>/dev/null
python $R/tools/kernel_stats_md.py $O/trace_bbc/t_kernel_stats.csv "rocprofv3 --kernel-trace --stats of python bench.py --workload bbc --steps 6 --warmup 2 (the 11-clip 640x360 stand-in behind the default downscale: 8 launches of the fused downscale kernel's SEG instance)" > $O/kernel_trace_bbc_flow.md 2>&1; head -8 $O/kernel_trace_bbc_flow.md | cut -c1-200; rm -rf $O/trace_bbc
ls $O | wc -l
