#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_k}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^(FAILED|ERROR)" $O/pytest.log | head -20; grep -E "^E " $O/pytest.log | head -20
for w in bbc corpus; do timeout 600 python bench.py --workload $w --steps 6 --warmup 3 > $O/bench_$w.json 2> $O/bench_$w.err; python - <<PY
import json
d=json.load(open("$O/bench_$w.json"))
print("$w", d["value"], "ms/step", d["ms_per_step"], "kernel ms", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d.get("parity_sample"), d.get("ground_truth"))
PY
done
timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 10 2>/dev/null | tail -1 | cut -c1-300
