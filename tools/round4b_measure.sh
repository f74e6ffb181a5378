#!/bin/bash
# Round-4 measurement matrix on one MI355X, second session (PMC passes only for the kernels that changed since r04_m: the
# all-detectors pass and the edge pipeline; the HSV pass again as the headline).  usage: tools/round4b_measure.sh <tag>
# default bench x 3 (+ median), flows, 1-rank launcher run, the secondary kernels on their own, kernel traces of the default
# bench and of the edge term on S / T / U content, PMC passes (HBM traffic) of the HSV pass, the fused pass, both fused
# downscale kernels, the hash thumbnails and the edge pipeline on frames with objects, host-feed rates, GPU test log.
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r04_ab}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R; export PYTHONPATH=$R:$R/tools
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 --timeout-method=thread > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for i in 1 2 3; do timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_default_$i.json; echo "bench rc=${PIPESTATUS[0]}"; done
python - <<PY
import json
rows=[json.load(open("$O/bench_default_%d.json" % i)) for i in (1,2,3)]
print("default bench value / frac / avg_launch_ms:", [(r["value"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"]) for r in rows])
r=sorted(rows, key=lambda r: r["roofline"]["avg_launch_ms"])[1]
json.dump(r, open("$O/bench_default_median.json","w"))
print("  parity:", r["parity_sample"][:200])
for k,v in (r.get("secondary") or {}).items():
    print("  ", k, {kk:vv for kk,vv in v.items() if kk in ("value","avg_launch_ms","frac_of_8TBps","error","ms_per_step")}, (v.get("roofline") or {}).get("frac"), (v.get("parity_sample") or "")[:60])
    for kk,vv in v.items():
        if isinstance(vv, dict) and "value" in vv: print("      ", kk, {a:b for a,b in vv.items() if a in ("value","avg_launch_ms","frac_of_8TBps","us_per_frame")})
print("  cpu", {k:v for k,v in r["cpu_baseline"].items() if k not in ("sample",)})
PY
for w in corpus bbc; do timeout 600 python bench.py --workload $w --steps 6 --warmup 3 2>/dev/null | tail -1 > $O/bench_$w.json; cut -c1-160 $O/bench_$w.json; done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 2>/dev/null | tail -1 > $O/bench_torchrun_1rank.json
timeout 300 python bench.py --downscale auto --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_downscale_auto.json
timeout 300 python bench.py --downscale auto --detector all --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_downscale_auto_all.json
timeout 300 python bench.py --detector all --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_all.json
timeout 300 python bench.py --detector hist --res 4k --frames 2048 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_hist_4k.json
timeout 300 python bench.py --dist S --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_content_S_4096.json
timeout 300 python bench.py --dist S --frames 2048 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_content_S_2048.json
for d in S T U; do timeout 300 python bench.py --detector edges --dist $d --frames 2048 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_edges_hsv_$d.json; done
for f in torchrun_1rank downscale_auto downscale_auto_all all hist_4k content_S_4096 content_S_2048 edges_hsv_S edges_hsv_T edges_hsv_U; do python -c "import json; d=json.load(open('$O/bench_$f.json')); print('$f', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done
timeout 300 python tools/feed_bench.py 2>/dev/null | tail -1 > $O/feed_bench.json; cut -c1-400 $O/feed_bench.json
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2>/dev/null
python $R/tools/kernel_stats_md.py $O/trace/t_kernel_stats.csv "rocprofv3 --kernel-trace --stats of python bench.py --no-cpu-baseline --no-secondary (headline: 23 launches of the HSV pass)" > $O/kernel_trace_default_bench.md 2>&1; head -8 $O/kernel_trace_default_bench.md | cut -c1-200; rm -rf $O/trace
cd $R; bash tools/edge_trace.sh gpurun_out/$T "" | grep -v "rocclr\|store_xor\|^$\|^|---"
cd /tmp
# PMC passes (separate runs per counter group, --kernel-trace only)
for what in content all; do
  case $what in content) BA="--frames 4096 --steps 2 --warmup 1"; K=score_frames;; all) BA="--frames 4096 --steps 2 --warmup 1 --detector all"; K=score_frames;;
    downscale) BA="--frames 4096 --steps 2 --warmup 1 --downscale auto"; K=resize_walk;; downscale_all) BA="--frames 4096 --steps 2 --warmup 1 --downscale auto --detector all"; K=resize_walk;; esac
  P=$O/pmc_$what; mkdir -p $P
  run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $P/$name -o pmc --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary $BA > $P/$name.log 2>&1; }
  run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
  run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  python $R/tools/pmc_summary.py $P $K > $O/pmc_$what.txt; cat $O/pmc_$what.txt
  rm -rf $P
done
for d in S T; do
  P=$O/pmc_edges_hsv_$d; mkdir -p $P
  runh() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $P/$name -o pmc --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 --detector edges --dist $d --frames 1024 > $P/$name.log 2>&1; }
  runh f FETCH_SIZE
  runh w WRITE_SIZE
  python $R/tools/pmc_by_kernel.py $P psd:: > $O/pmc_edges_hsv_traffic_$d.txt; cut -c1-230 $O/pmc_edges_hsv_traffic_$d.txt
  rm -rf $P
done
ls $O | head -80
