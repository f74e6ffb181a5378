#!/bin/bash
# Round-3 measurement matrix on one MI355X.  usage: tools/round3_measure.sh <tag>   (writes gpurun_out/<tag>/)
# = tools/round_measure.sh (round 2) + the two flow workloads (BASELINE configs 4 and 5) + HBM traffic counters of the edge
# pipeline and the fused all-detectors pass + kernel traces of the edges + HSV workload.
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_m}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 --timeout-method=thread > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -2 $O/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for i in 1 2 3; do timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_default_$i.json; done
python - <<PY
import json
rows=[json.load(open("$O/bench_default_%d.json" % i)) for i in (1,2,3)]
print("default bench value / frac / avg_launch_ms:", [(r["value"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"]) for r in rows])
r=sorted(rows, key=lambda r: r["roofline"]["avg_launch_ms"])[1]
json.dump(r, open("$O/bench_default_median.json","w"))
for k,v in (r.get("secondary") or {}).items():
    print("  ", k, {kk:vv for kk,vv in v.items() if kk in ("value","avg_launch_ms","frac_of_8TBps","error","ms_per_step")}, (v.get("roofline") or {}).get("frac"))
    for kk,vv in v.items():
        if isinstance(vv, dict) and "value" in vv: print("      ", kk, {a:b for a,b in vv.items() if a in ("value","avg_launch_ms","frac_of_8TBps")})
print("  cpu", {k:v for k,v in r["cpu_baseline"].items() if k not in ("sample",)})
PY
for w in corpus bbc; do timeout 600 python bench.py --workload $w --steps 6 --warmup 3 2>/dev/null | tail -1 > $O/bench_$w.json; cut -c1-160 $O/bench_$w.json; done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 2>/dev/null | tail -1 > $O/bench_torchrun_1rank.json
timeout 300 python bench.py --downscale auto --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_downscale_auto.json
timeout 300 python bench.py --detector all --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_all.json
timeout 300 python bench.py --detector edges --dist S --frames 2048 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_edges_hsv_S.json
timeout 300 python bench.py --dist S --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_content_S_4096.json
for f in torchrun_1rank downscale_auto all edges_hsv_S content_S_4096; do python -c "import json; d=json.load(open('$O/bench_$f.json')); print('$f', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done
timeout 300 python tools/feed_bench.py 2>/dev/null | tail -1 > $O/feed_bench.json
( for n in 64 256 1024; do ET_N=$n ET_SMOOTH=1 timeout 200 python tools/edge_time.py 2>/dev/null | tail -1 | sed "s/^/shot-like /"; done; ET_N=256 timeout 200 python tools/edge_time.py 2>/dev/null | tail -1 | sed "s/^/uniform noise /" ) | tee $O/edge_time.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2>/dev/null
python $R/tools/kernel_stats_md.py $O/trace/t_kernel_stats.csv "rocprofv3 --kernel-trace --stats of python bench.py --no-cpu-baseline --no-secondary (headline: 23 launches of the HSV pass)" > $O/kernel_trace_default_bench.md 2>&1; head -8 $O/kernel_trace_default_bench.md | cut -c1-200; rm -rf $O/trace
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 1 --detector edges --dist S --frames 2048 > /dev/null 2>&1
python $R/tools/kernel_stats_md.py $O/trace/t_kernel_stats.csv "rocprofv3 --kernel-trace --stats of bench.py --detector edges --dist S --frames 2048 --steps 4 --warmup 1 (edges + HSV from one read, 5 steps of 2048 x 1080p shot-like frames)" > $O/kernel_trace_edges_hsv_S.md 2>&1; head -14 $O/kernel_trace_edges_hsv_S.md | cut -c1-200; rm -rf $O/trace
( ET_N=256 ET_SMOOTH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/tools/edge_time.py > /dev/null 2>&1 )
python $R/tools/kernel_stats_md.py $O/trace/t_kernel_stats.csv "rocprofv3 --kernel-trace --stats of ET_N=256 ET_SMOOTH=1 tools/edge_time.py (edge term alone, 5 calls of 256 x 1080p shot-like frames)" > $O/edge_pipeline_kernel_trace_shotlike_N256.md 2>&1; head -12 $O/edge_pipeline_kernel_trace_shotlike_N256.md | cut -c1-160; rm -rf $O/trace
# PMC passes (separate runs per counter group, --kernel-trace only): HSV pass, fused pass, downscale kernel, edge pipeline
for what in content all downscale; do
  case $what in content) BA="--frames 4096 --steps 2 --warmup 1"; K=score_frames;; all) BA="--frames 4096 --steps 2 --warmup 1 --detector all"; K=score_frames;; downscale) BA="--frames 4096 --steps 2 --warmup 1 --downscale auto"; K=resize_walk;; esac
  P=$O/pmc_$what; mkdir -p $P
  run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $P/$name -o pmc --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary $BA > $P/$name.log 2>&1; }
  run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
  run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  python $R/tools/pmc_summary.py $P $K > $O/pmc_$what.txt; cat $O/pmc_$what.txt
  rm -rf $P
done
P=$O/pmc_edges; mkdir -p $P
rune() { name=$1; shift; ET_N=256 ET_SMOOTH=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $P/$name -o pmc --output-format csv -- python $R/tools/edge_time.py > $P/$name.log 2>&1; }
rune f FETCH_SIZE
rune w WRITE_SIZE
python $R/tools/pmc_by_kernel.py $P psd:: > $O/pmc_edges_traffic.txt; cut -c1-230 $O/pmc_edges_traffic.txt
rm -rf $P
P=$O/pmc_edges_hsv; mkdir -p $P
runh() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $P/$name -o pmc --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 --detector edges --dist S --frames 2048 > $P/$name.log 2>&1; }
runh f FETCH_SIZE
runh w WRITE_SIZE
python $R/tools/pmc_by_kernel.py $P psd:: > $O/pmc_edges_hsv_traffic.txt; cut -c1-230 $O/pmc_edges_hsv_traffic.txt
rm -rf $P
ls $O | head -60
