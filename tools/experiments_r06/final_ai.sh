cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_ai; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for i in 1 2 3; do timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_default_$i.json; done
for i in 1 2 3; do timeout 300 python bench.py --workload corpus --steps 6 --warmup 3 2>/dev/null | tail -1 > $O/bench_corpus_$i.json; done
timeout 300 python bench.py --workload bbc --steps 6 --warmup 3 2>/dev/null | tail -1 > $O/bench_bbc.json
python - <<PY
import json
O="$O/"
for i in (1,2,3):
    d=json.load(open(O+"bench_default_%d.json"%i)); s=d["secondary"]
    print("default",i,d["value"],d["roofline"]["avg_launch_ms"],d["roofline"]["frac"], "corpus sec", s["corpus_mixed_1080p_4k_all_four"]["value"], s["corpus_mixed_1080p_4k_all_four"]["ms_per_step"], "bbc sec", s["bbc_standin_adaptive"]["value"], "all4 ds", s["default_pipeline_downscale_auto_all_four"]["frac_of_8TBps"], "hist4k", s["histogram_threshold_4k"]["frac_of_8TBps"])
for i in (1,2,3):
    d=json.load(open(O+"bench_corpus_%d.json"%i)); print("corpus",i,d["value"],d["ms_per_step"],d["roofline"]["avg_launch_ms"],d["roofline"]["frac"], d["parity_sample"][:90])
d=json.load(open(O+"bench_bbc.json")); print("bbc",d["value"],d["ms_per_step"],d["roofline"]["avg_launch_ms"],d["roofline"]["frac"])
PY
