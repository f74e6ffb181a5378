#!/bin/bash
# round 5, second probe: the flows with pipelined pieces (A/B over PSD_CLIPS_SPLIT_MB), new GPU tests, the default bench
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
timeout 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_flows.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
for mb in 0 2048 4096 8192; do
  for w in bbc corpus; do
    PSD_CLIPS_SPLIT_MB=$mb timeout 300 python bench.py --workload $w --bbc-frames 2000 --corpus-frames 512 --steps 5 --warmup 2 --cpu-sample 64 > $O/flow_small_${w}_$mb.json 2> $O/flow_small_${w}_$mb.err
    python -c "import json; d=json.load(open('$O/flow_small_${w}_$mb.json')); print('small $w split=$mb', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['parity_sample'][:40])"
  done
done
for mb in 0 4096 8192; do
  for w in bbc corpus; do
    PSD_CLIPS_SPLIT_MB=$mb timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --cpu-sample 64 > $O/flow_full_${w}_$mb.json 2> $O/flow_full_${w}_$mb.err
    python -c "import json; d=json.load(open('$O/flow_full_${w}_$mb.json')); print('full $w split=$mb', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['parity_sample'][:40])"
  done
done
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_default.json")); print(d["value"], d["roofline"]["frac"])
for k,v in d["secondary"].items():
    print(k, v.get("value"), v.get("frac_of_8TBps") or (v.get("roofline") or {}).get("frac"), v.get("with_stats_manager_frames_per_s"), v.get("ms_per_step"), (v.get("parity_sample") or v.get("error") or "")[:50])
PY
