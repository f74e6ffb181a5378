#!/bin/bash
# End-of-round measurement matrix on one MI355X.  usage: tools/round_measure.sh <tag>   (writes gpurun_out/<tag>/)
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-rXX}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout=600 --timeout-method=thread > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -2 $O/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# the driver's line: default flags (headline + secondary + CPU baselines), three runs for the median
for i in 1 2 3; do timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_default_$i.json; done
python - <<PY
import json
rows=[json.load(open("$O/bench_default_%d.json" % i)) for i in (1,2,3)]
print("default bench value / frac / avg_launch_ms:", [(r["value"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"]) for r in rows])
r=sorted(rows, key=lambda r: r["roofline"]["avg_launch_ms"])[1]
json.dump(r, open("$O/bench_default_median.json","w"))
for k,v in (r.get("secondary") or {}).items(): print("  ", k, {kk:vv for kk,vv in v.items() if kk in ("value","avg_launch_ms","frac_of_8TBps","error")})
print("  cpu", {k:v for k,v in r["cpu_baseline"].items() if k not in ("sample",)})
PY
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 2>/dev/null | tail -1 > $O/bench_torchrun_1rank.json
timeout 300 python bench.py --downscale auto --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_downscale_auto.json
timeout 300 python bench.py --detector all --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_all.json
timeout 300 python tools/feed_bench.py 2>/dev/null | tail -1 > $O/feed_bench.json
( for n in 64 256 1024; do ET_N=$n ET_SMOOTH=1 timeout 200 python tools/edge_time.py 2>/dev/null | tail -1 | sed "s/^/shot-like /"; done; ET_N=256 timeout 200 python tools/edge_time.py 2>/dev/null | tail -1 | sed "s/^/uniform noise /" ) > $O/edge_time.txt
timeout 300 python tools/bbc_standin.py --dump $O/bbc_standin_predictions.json 2>/dev/null | tail -1 > $O/bbc_standin.json
# kernel trace of the default bench command (headline only), then PMC passes of the HSV pass and of the downscale kernel
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2>/dev/null
cd $R; python tools/rocpd_summary.py $O/trace/*.db > $O/kernel_trace_default_bench.md 2>&1; grep -v "at::native\|rocclr" $O/kernel_trace_default_bench.md | cut -c1-220 | head -8
( cd /tmp; ET_N=256 ET_SMOOTH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/edge_trace -o t --output-format csv -- python $R/tools/edge_time.py > /dev/null 2>&1 )
python tools/kernel_stats_md.py $O/edge_trace/t_kernel_stats.csv "rocprofv3 --kernel-trace --stats of ET_N=256 ET_SMOOTH=1 tools/edge_time.py (edge term alone, 3 calls of 256 x 1080p shot-like frames)" > $O/edge_pipeline_kernel_trace_shotlike_N256.md 2>&1; head -12 $O/edge_pipeline_kernel_trace_shotlike_N256.md | cut -c1-160
rm -rf $O/edge_trace
for what in content downscale; do
  if [ $what = content ]; then BA="--frames 4096 --steps 2 --warmup 1"; K=score_frames; else BA="--frames 4096 --steps 2 --warmup 1 --downscale auto"; K=resize_walk; fi
  P=$O/pmc_$what; mkdir -p $P
  ( cd /tmp
  run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $P/$name -o pmc --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary $BA > $P/$name.log 2>&1; }
  run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
  run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum )
  python tools/pmc_summary.py $P $K > $O/pmc_$what.txt; cat $O/pmc_$what.txt
  rm -rf $P/*/pmc_agent_info.csv
done
ls $O | head -40
