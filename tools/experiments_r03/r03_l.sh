#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_l}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^(FAILED|ERROR)" $O/pytest.log | head -20; grep -E "^E " $O/pytest.log | head -20
( for n in 64 256 1024; do ET_N=$n ET_SMOOTH=1 timeout 200 python tools/edge_time.py 2>/dev/null | tail -1 | sed "s/^/shot-like /"; done; ET_N=256 timeout 200 python tools/edge_time.py 2>/dev/null | tail -1 | sed "s/^/uniform noise /" ) | tee $O/edge_time.txt
for w in corpus bbc; do timeout 600 python bench.py --workload $w --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; python - <<PY
import json
d=json.load(open("$O/bench_$w.json"))
print("$w", d["value"], "ms/step", d["ms_per_step"], "kernel ms", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d.get("parity_sample"))
PY
done
timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 8 --detector edges --dist S --frames 2048 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('edges+HSV S 2048', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
